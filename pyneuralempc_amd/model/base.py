"""Model plug-in interface (reference: pyNeuralEMPC/model/base.py:3-18).

A Model maps H rows of (state, control [, parameters]) to H rows of the state-sized network output and supplies the
first and second derivatives of that map in the reference's block layout.  The call-site signature is
``forward / jacobian / hessian(x, u, p=None, tvp=None)`` -- what the reference's integrators actually pass
(integrator/discret.py:27,48,64); the extra positional ``x0`` of the reference's abstract base is never supplied by any
caller and is not reproduced.  Subclasses that run on the device derive from ``MLPModel`` (model/mlp.py)."""

# variable kinds (kept for scripts that import them; unused by the solver path, as in the reference)
CONSTANT_VAR, CONTROL_VAR, STATE_VAR = 1, 2, 3


def _plugin_method(what):
    def method(self, x, u, p=None, tvp=None):
        raise NotImplementedError("")
    method.__name__ = what
    method.__doc__ = f"{what}(x (H,x_dim), u (H,u_dim), p=None, tvp=None): provided by the concrete model."
    return method


class Model:
    """Dimensions of the plug-in: state, control, constant parameters, time-varying parameters."""

    forward = _plugin_method("forward")      # (H, x_dim)
    jacobian = _plugin_method("jacobian")    # (H*x_dim, H*x_dim + H*u_dim), columns [all x | all u]
    hessian = _plugin_method("hessian")      # (H, x_dim, n, n) with n = H*(x_dim + u_dim)

    def __init__(self, x_dim: int, u_dim: int, p_dim=None, tvp_dim=None):
        self.x_dim, self.u_dim = x_dim, u_dim
        self.p_dim, self.tvp_dim = p_dim, tvp_dim


class ReOrderProxyModel(Model):
    """Declared and unimplemented in the reference as well (model/base.py:26-28)."""

    def __init__(self, model, order_list: list):
        raise NotImplementedError("")
