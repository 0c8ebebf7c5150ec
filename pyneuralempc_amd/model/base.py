"""Model plug-in interface (reference: pyNeuralEMPC/model/base.py:3-18).

Call-site signature is forward/jacobian/hessian(x, u, p=None, tvp=None) -- what the reference's
integrators actually pass (integrator/discret.py:27,48,64); the extra positional ``x0`` of the
reference's abstract base is never supplied by any caller and is not reproduced."""


class Model:
    def __init__(self, x_dim: int, u_dim: int, p_dim=None, tvp_dim=None):
        self.x_dim = x_dim
        self.u_dim = u_dim
        self.p_dim = p_dim
        self.tvp_dim = tvp_dim

    def forward(self, x, u, p=None, tvp=None):
        raise NotImplementedError("")

    def jacobian(self, x, u, p=None, tvp=None):
        raise NotImplementedError("")

    def hessian(self, x, u, p=None, tvp=None):
        raise NotImplementedError("")


CONSTANT_VAR = 1
CONTROL_VAR = 2
STATE_VAR = 3
