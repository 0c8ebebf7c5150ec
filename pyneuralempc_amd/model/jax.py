"""Names kept for scripts written against the reference (model/jax.py:32,93): both classes wrap an arbitrary JAX
callable there.  The device path evaluates feed-forward tanh networks given by their weights; constructing one of these
explains the replacement instead of failing with an AttributeError."""
from .base import Model

_MSG = ("{name} wraps a JAX callable, which cannot be compiled into the HIP kernels.  Give the dynamics as a feed-forward "
        "tanh network: model.MLPModel(weights, biases, x_dim, u_dim) / model.tensorflow.KerasTFModel(keras_model, ...), "
        "or model.MLPModelRollingInput(..., rolling_window=w) for the rolling-window variant.")


class DiffDiscretJaxModel(Model):
    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, vector_mode=False, safe_mode=True):
        raise NotImplementedError(_MSG.format(name="DiffDiscretJaxModel"))


class DiffDiscretJaxModelRollingWindow(Model):
    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, rolling_window=1, forward_rolling=True,
                 vector_mode=True, safe_mode=True):
        raise NotImplementedError(_MSG.format(name="DiffDiscretJaxModelRollingWindow"))
