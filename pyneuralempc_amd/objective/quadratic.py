"""Quadratic + linear stage cost evaluated on the device.

Stands in for JAXObjectifFunc (reference: objective/jax.py:16-90), which differentiates an
arbitrary JAX callable on the CPU.  An arbitrary Python callable cannot be compiled to a HIP
kernel; the family below covers both objectives the reference's own scripts use
(examples/lotka_volterra/run.py:79-84  sum(u * c);  test.py:55-60  sum((u - 2)^2)):

    f = sum_t (x_t - xref_t)^T Q (x_t - xref_t) + (u_t - uref_t)^T R (u_t - uref_t)
              + cx_t . x_t + cu_t . u_t

with an optional terminal weight QT replacing Q in the last step (t = H-1).
"""
import numpy as np
import torch

from .base import ObjectiveFunc
from ..engine import CallbackEngine


class QuadraticObjective(ObjectiveFunc):
    def __init__(self, Q=None, R=None, xref=None, uref=None, cx=None, cu=None, QT=None, dtype=torch.float64,
                 device="cuda"):
        super().__init__()
        self.params = dict(Q=Q, R=R, xref=xref, uref=uref, cx=cx, cu=cu, QT=QT)
        self.dtype, self.device = dtype, device
        self._engines = {}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_engines"] = {}
        return d

    def resolved(self, H, nx, nu):
        """Parameters with defaults filled in (Q = I, R = 0.1 I: SURVEY.md 8d)."""
        p = self.params
        Q = np.eye(nx) if p["Q"] is None else np.asarray(p["Q"], dtype=np.float64).reshape(nx, nx)
        R = 0.1 * np.eye(nu) if p["R"] is None else np.asarray(p["R"], dtype=np.float64).reshape(nu, nu)

        def tv(v, d):
            return np.zeros((H, d)) if v is None else np.broadcast_to(np.asarray(v, dtype=np.float64), (H, d)).copy()
        QT = None if p.get("QT") is None else np.asarray(p["QT"], dtype=np.float64).reshape(nx, nx)
        return dict(Q=Q, R=R, xref=tv(p["xref"], nx), uref=tv(p["uref"], nu), cx=tv(p["cx"], nx), cu=tv(p["cu"], nu),
                    QT=QT)

    def fingerprint(self, H, nx, nu):
        """Bytes of the resolved parameters: what a device handle compares to notice that `params` was changed after
        the objective was uploaded (a moving xref / uref in tracking MPC)."""
        r = self.resolved(H, nx, nu)
        return b"".join(np.ascontiguousarray(r[k]).tobytes() if r[k] is not None else b"-"
                        for k in ("Q", "R", "xref", "uref", "cx", "cu", "QT"))

    def _engine(self, H, nx, nu):
        key = (H, nx, nu)
        ent = self._engines.get(key)
        fp = self.fingerprint(H, nx, nu)
        if ent is None:
            # objective-only handle: a one-layer placeholder network (never evaluated)
            eng = CallbackEngine([np.zeros((nx + nu, nx))], [np.zeros(nx)], H, nx, nu, integrator="unity",
                                 dtype=self.dtype, device=self.device, max_batch=1, kernel="valu")
            eng.set_objective(**self.resolved(H, nx, nu))
            self._engines[key] = [eng, fp]
            return eng
        if ent[1] != fp:            # parameters were edited since the upload: the reference re-reads them every call
            ent[0].set_objective(**self.resolved(H, nx, nu))
            ent[1] = fp
        return ent[0]

    def _z(self, states, u):
        states, u = np.asarray(states, dtype=np.float64), np.asarray(u, dtype=np.float64)
        eng = self._engine(states.shape[0], states.shape[1], u.shape[1])
        return eng, eng.to_device(np.concatenate([states.reshape(-1), u.reshape(-1)])[None, :])

    def forward(self, states, u, p=None, tvp=None):
        eng, Z = self._z(states, u)
        return float(eng.eval(Z, Z[:, :eng.nx].contiguous(), want=("f",))["f"][0].item())

    def gradient(self, states, u, p=None, tvp=None):
        eng, Z = self._z(states, u)
        return eng.eval(Z, Z[:, :eng.nx].contiguous(), want=("grad",))["grad"][0].to("cpu", torch.float64).numpy()

    def hessian(self, states, u, p=None, tvp=None):
        """Constant block-diagonal (n,n): Q+Q^T on every x_t block, R+R^T on every u_t block."""
        states, u = np.asarray(states), np.asarray(u)
        H, nx, nu = states.shape[0], states.shape[1], u.shape[1]
        r = self.resolved(H, nx, nu)
        n = H * (nx + nu)
        Hm = np.zeros((n, n))
        for t in range(H):
            Qt = r["QT"] if (t == H - 1 and r["QT"] is not None) else r["Q"]
            Hm[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = Qt + Qt.T
            o = H * nx + t * nu
            Hm[o:o + nu, o:o + nu] = r["R"] + r["R"].T
        return Hm

    def hessianstructure(self, H, model):
        nx, nu = model.x_dim, model.u_dim
        n = H * (nx + nu)
        M = np.zeros((n, n))
        for t in range(H):
            M[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = 1.0
            o = H * nx + t * nu
            M[o:o + nu, o:o + nu] = 1.0
        return M
