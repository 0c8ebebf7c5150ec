"""Names kept for scripts written against the reference (model/jax.py:32,93): both classes wrap an arbitrary JAX
callable there.  JAX is not part of this build.  The same contract -- a differentiable function of the whole trajectory,
differentiated by autodiff, results in the reference's block layouts -- is offered for a torch callable by
``model.TorchModel`` / ``model.TorchModelRollingWindow`` (torch.func on the device, through the integrators' host algebra and
the unfused solver glue); dense
networks given by their weights run inside the HIP kernels (``MLPModel`` / ``KerasTFModel`` / ``MLPModelRollingInput``).
Constructing one of the JAX classes explains the replacement instead of failing with an AttributeError."""
from .base import Model

_MSG = ("{name} wraps a JAX callable; JAX is not part of this build.  Write the function with torch operations and wrap it "
        "in model.TorchModel(forward_func, x_dim, u_dim, p_dim, tvp_dim, vector_mode=True) -- same signature "
        "forward_func(x, u, p=None, tvp=None), same forward / jacobian / hessian layouts (model/jax.py:45-88), differentiated "
        "with torch.func.  A dense network given by its weights runs in the HIP kernels instead: model.MLPModel(weights, "
        "biases, x_dim, u_dim, activations=...) / model.tensorflow.KerasTFModel(keras_model, ...){rolling}.")


class DiffDiscretJaxModel(Model):
    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, vector_mode=False, safe_mode=True):
        raise NotImplementedError(_MSG.format(name="DiffDiscretJaxModel", rolling=""))


class DiffDiscretJaxModelRollingWindow(Model):
    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, rolling_window=1, forward_rolling=True,
                 vector_mode=True, safe_mode=True):
        raise NotImplementedError(_MSG.format(
            name="DiffDiscretJaxModelRollingWindow",
            rolling="; the rolling-window form of the callable is model.TorchModelRollingWindow(forward_func, x_dim, u_dim, "
                    "p_dim, tvp_dim, rolling_window=w) with the same set_prev_data, a rolling-window network given by its "
                    "weights model.MLPModelRollingInput(..., rolling_window=w)"))
