from .base import DeviceIntegrator


class RK4Integrator(DeviceIntegrator):
    """Classic RK4 step with the control held over the step (reference: integrator/rk4.py:46-285).

    ``cache_mode`` / ``cache_size`` are accepted for signature compatibility and ignored: the
    reference's TensorCache (rk4.py:20-43) shares stage values between the forward / jacobian /
    hessian callbacks of one iterate; here one fused launch produces them together."""
    KIND = "rk4"

    def __init__(self, model, H, DT, cache_mode=False, cache_size=2):
        if int(getattr(model, "rolling_window", 1)) > 1:
            # same limit as the reference: its rolling models only pair with DiscretIntegrator / UnityIntegrator
            raise NotImplementedError("RK4Integrator does not support rolling-window models")
        super().__init__(model, H, DT=DT)
        self.cache_mode = cache_mode
