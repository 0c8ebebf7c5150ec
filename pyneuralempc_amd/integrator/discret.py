from .base import DeviceIntegrator


class DiscretIntegrator(DeviceIntegrator):
    """x_t = x_{t-1} + f(x_{t-1}, u_t)   (reference: integrator/discret.py:7-81)."""
    KIND = "discret"

    def __init__(self, model, H):
        super().__init__(model, H)
