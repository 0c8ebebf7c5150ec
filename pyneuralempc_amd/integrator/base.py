"""Integrator plug-in interface (reference: pyNeuralEMPC/integrator/base.py:5-123).

An Integrator turns a one-step Model into H*x_dim defect constraints  Phi(x_{t-1},u_t) - x_t = 0
over the decision vector z = [states (H,nx) | controls (H,nu)] and supplies their Jacobian and
Hessian.  In this build the three concrete integrators evaluate everything on the device through
one CallbackEngine per integrator; the per-problem NumPy methods below keep the reference's
signatures, the ``*_batch`` twins take (B, ...) device tensors."""
import numpy as np
import torch

from ..model.base import Model
from ..model.mlp import MLPModel


class Integrator:
    KIND = None  # "discret" | "unity" | "rk4" for the device-backed subclasses

    def __init__(self, model, H: int, nb_contraints: int):
        if not isinstance(model, (Model,)):
            raise ValueError("The model provided isn't a Model object !")
        self.H = H
        self.model = model
        self.nb_contraints = nb_contraints
        self.hessian_structure_cache = None
        self._engine = None
        self._fused = {}

    def get_dim(self):
        raise NotImplementedError("")

    def get_bound(self):
        raise NotImplementedError("")

    def forward(self, x, u, x0, p=None, tvp=None):
        raise NotImplementedError("")

    def jacobian(self, x, u, x0, p=None, tvp=None):
        raise NotImplementedError("")

    def hessian(self, x, u, x0, p=None, tvp=None):
        raise NotImplementedError("")

    def hessianstructure(self):
        if self.hessian_structure_cache is None:
            self.hessian_structure_cache = self._compute_hessianstructure()
        return self.hessian_structure_cache

    def _compute_hessianstructure(self):
        raise NotImplementedError("")

    def get_lower_bounds(self, _):
        return [0.0, ] * self.nb_contraints

    def get_upper_bounds(self, _):
        return [0.0, ] * self.nb_contraints

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_engine"] = None
        d["_fused"] = {}
        return d


class DeviceIntegrator(Integrator):
    """Shared implementation of Discret / Unity / RK4: on the HIP engine for dense-network models (MLPModel /
    KerasTFModel and their rolling-window forms), through the generic host algebra (integrator/host.py) for every other
    Model plug-in (model.TorchModel, a user's own subclass) -- those have no kernel, exactly as in the reference every
    model goes through Model.forward / jacobian / hessian."""

    def __init__(self, model, H, DT=1.0):
        if not isinstance(model, Model):
            raise ValueError("The model provided isn't a Model object !")
        super().__init__(model, H, model.x_dim * H)
        self.DT = DT
        self.on_device = isinstance(model, MLPModel)
        self._host = None
        if not self.on_device:
            from .host import HostAlgebra
            # a rolling-window callable (model.TorchModelRollingWindow): its block-layout Jacobian / Hessian are banded over
            # the window and the index map of the host algebra places any band; the RK4 stage recursion is per-row only
            if int(getattr(model, "rolling_window", 1)) > 1 and self.KIND == "rk4":
                raise NotImplementedError("rolling-window models need the Discret or Unity integrator")
            self._host = HostAlgebra(model, H, self.KIND, DT)

    def _need_device(self, what):
        if not self.on_device:
            raise NotImplementedError(f"{what} needs a dense-network model (MLPModel / KerasTFModel): "
                                      f"{type(self.model).__name__} is evaluated on the host, one problem per call")

    # -- engine ---------------------------------------------------------------------------
    def engine(self, max_batch=1):
        self._need_device("the batched device engine")
        if self._engine is None:
            self._engine = self.model.make_engine(self.H, self.KIND, DT=self.DT, max_batch=max_batch)
        self._engine.reserve(max_batch)
        return self._engine

    def _pack(self, x, u, x0, p=None, tvp=None):
        assert len(x.shape) == 2 and len(u.shape) == 2, "x and u tensor must have dim 2"
        assert len(x0.shape) == 1, "x0 shape must have dim 1"
        eng = self.engine(1)
        self.model.bind_inputs(eng, p, tvp)
        z = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(u, dtype=np.float64).reshape(-1)])
        return eng, eng.to_device(z[None, :]), eng.to_device(np.asarray(x0, dtype=np.float64)[None, :])

    # -- reference signatures (one problem, NumPy in / out) ---------------------------------
    def forward(self, x, u, x0, p=None, tvp=None):
        if self._host is not None:
            return self._host.forward(x, u, x0, p=p, tvp=tvp)
        eng, Z, X0 = self._pack(x, u, x0, p, tvp)
        g = eng.eval(Z, X0, want=("g",))["g"]
        return g[0, :self.nb_contraints].to("cpu", torch.float64).numpy()

    def jacobian(self, x, u, x0, p=None, tvp=None):
        if self._host is not None:
            return self._host.jacobian(x, u, x0, p=p, tvp=tvp)
        eng, Z, X0 = self._pack(x, u, x0, p, tvp)
        J = eng.eval(Z, X0, want=("jac_dense",))["jac_dense"]
        return J[0, :self.nb_contraints].to("cpu", torch.float64).numpy()

    def hessian(self, x, u, x0, p=None, tvp=None):
        """(H*nx, n, n) like integrator/discret.py:61-81: one device call per constraint row block
        (one-hot multipliers); meant for inspection, the solver path uses the contracted form."""
        if self._host is not None:
            return self._host.hessian(x, u, x0, p=p, tvp=tvp)
        eng, Z, X0 = self._pack(x, u, x0, p, tvp)
        n, m = eng.n, eng.m
        out = np.zeros((self.nb_contraints, n, n))
        sigma = torch.zeros(1, dtype=eng.dtype, device=eng.device)
        for i in range(self.nb_contraints):
            lam = torch.zeros(1, m, dtype=eng.dtype, device=eng.device)
            lam[0, i] = 1.0
            out[i] = eng.hess(Z, X0, lam, sigma, want=("hdense",))["hdense"][0].to("cpu", torch.float64).numpy()
        return out

    def _compute_hessianstructure(self):
        """Exact band pattern as an (n,n) 0/1 float map (the reference ORs three random samples,
        integrator/base.py:89-115)."""
        H, nx, nu = self.H, self.model.x_dim, self.model.u_dim
        n = H * (nx + nu)
        M = np.zeros((n, n))
        w = int(getattr(self.model, "rolling_window", 1))
        fwd = bool(getattr(self.model, "forward_rolling", True))
        for t in range(H):
            # variables step t reads: states x_{t-w} .. x_{t-1} (x0 and the history are data) and controls
            # u_{t-w+1} .. u_t; every pair of them can couple
            tau = t + (np.arange(w) - (w - 1) if fwd else -np.arange(w))
            cols = [np.arange((k - 1) * nx, k * nx) for k in tau if k >= 1]
            cols += [np.arange(H * nx + k * nu, H * nx + (k + 1) * nu) for k in tau if k >= 0]
            cols = np.concatenate(cols)
            M[np.ix_(cols, cols)] = 1.0
        return M

    # -- batched twins (device tensors) -------------------------------------------------------
    def forward_batch(self, Z, X0):
        return self.engine(Z.shape[0]).eval(Z, X0, want=("g",))["g"][:, :self.nb_contraints]

    def jacobian_batch(self, Z, X0, layout="dense"):
        key = {"dense": "jac_dense", "tiles": "jac_tiles", "sparse": "jac_sparse"}[layout]
        return self.engine(Z.shape[0]).eval(Z, X0, want=(key,))[key]
