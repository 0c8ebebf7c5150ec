from .base import DeviceIntegrator


class UnityIntegrator(DeviceIntegrator):
    """x_t = f(x_{t-1}, u_t)   (reference: integrator/unity.py:9-81)."""
    KIND = "unity"

    def __init__(self, model, H):
        super().__init__(model, H)
