"""Feed-forward dense network evaluated on the device (reference: model/tensorflow.py:8-109,
model/jax.py:32-88 -- there the derivatives come from TF / JAX autodiff of the whole (H, .) batch,
here from the analytic per-row sweeps in csrc/)."""
import numpy as np
import torch

from .base import Model
from ..engine import CallbackEngine


class MLPModel(Model):
    """x_{next-ish} = f([x | u]) with f a stack of Dense layers (Dense-tanh ... Dense-linear by default).

    weights[l] is the Keras ``kernel`` (in, out), biases[l] is (out,).  The output width must be
    x_dim and the input width x_dim + u_dim (same checks as KerasTFModel.__init__).
    activations: None (tanh hidden layers, linear output), one Keras activation name for every hidden layer, or one name
    per layer including the output layer -- linear, tanh, relu, sigmoid, softplus, elu.  One non-linear activation on
    all hidden layers with a linear output runs on the matrix-core kernels; any other mix on the generic kernel."""

    def __init__(self, weights, biases, x_dim, u_dim, p_dim=0, tvp_dim=0, dtype=torch.float64, device="cuda",
                 kernel="auto", _input_width=None, activations=None):
        p_dim, tvp_dim = int(p_dim or 0), int(tvp_dim or 0)
        weights = [np.asarray(w, dtype=np.float64) for w in weights]
        biases = [np.asarray(b, dtype=np.float64).reshape(-1) for b in biases]
        if len(weights) == 0 or len(weights) != len(biases):
            raise ValueError("weights and biases must be non-empty lists of equal length")
        if weights[-1].shape[1] != x_dim:
            raise ValueError("Your model do not provide a suitable output dim ! \n It must get the same dim as "
                             "the state dim.")
        if weights[0].shape[0] != (x_dim + u_dim + p_dim + tvp_dim if _input_width is None else _input_width):
            raise ValueError("Your model do not provide a suitable input dim ! \n It must get the same dim as the "
                             "sum of all input vars (x, u, p, tvp).")
        super().__init__(x_dim, u_dim, p_dim, tvp_dim)
        self.weights, self.biases = weights, biases
        from ..engine import resolve_activations
        self.activations = resolve_activations(activations, len(weights))
        self.dtype, self.device, self.kernel = dtype, device, kernel
        self._row_engine = None

    # device handles never travel through pickle (the reference drops its Keras model the same way,
    # model/tensorflow.py:31-37)
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_row_engine"] = None
        return d

    def make_engine(self, H, integrator, DT=1.0, max_batch=1):
        return CallbackEngine(self.weights, self.biases, H, self.x_dim, self.u_dim, integrator=integrator, DT=DT,
                              dtype=self.dtype, device=self.device, max_batch=max_batch, kernel=self.kernel,
                              n_extra=self.p_dim + self.tvp_dim, activations=self.activations)

    def gather_extra(self, rows, p=None, tvp=None):
        """(rows, tvp_dim + p_dim) array [tvp_t | p] in the reference's concatenation order
        (KerasTFModel._gather_input, model/tensorflow.py:39-47; the constant p is repeated on every row, which is
        what that method intends -- its own p branch builds a 3-D array and fails)."""
        if self.p_dim + self.tvp_dim == 0:
            return None
        parts = []
        if self.tvp_dim:
            if tvp is None:
                raise ValueError("this model has tvp_dim > 0: pass tvp (H, tvp_dim)")
            tvp = np.asarray(tvp, dtype=np.float64)
            assert tvp.shape == (rows, self.tvp_dim), "tvp first dim must set according to the horizon size !"
            parts.append(tvp)
        if self.p_dim:
            if p is None:
                raise ValueError("this model has p_dim > 0: pass p (p_dim,)")
            parts.append(np.tile(np.asarray(p, dtype=np.float64).reshape(1, self.p_dim), (rows, 1)))
        return np.concatenate(parts, axis=1)

    def bind_inputs(self, eng, p=None, tvp=None):
        """Bind everything one problem's evaluation reads besides (z, x0) to `eng` (batch of one): the extra network
        inputs here, plus the history for rolling-window models.  Returns True when something was (re)bound."""
        ex = self.gather_extra(eng.H, p, tvp)
        if ex is None:
            return False
        eng.bind_extra(eng.to_device(ex[None]))
        return True

    def _rows(self, R):
        # H=1 UNITY problem per row: x_prev = X0[r], u = Z[r, nx:], and with the state slot of Z
        # zero the defect Phi - 0 is exactly f(x, u)
        if self._row_engine is None:
            self._row_engine = self.make_engine(1, "unity", max_batch=max(R, 1))
        self._row_engine.reserve(R)
        return self._row_engine

    def rows_batch(self, X, U, want_jac=True, E=None):
        """Device API: X (R,nx), U (R,nu) [, E (R,tvp+p)] tensors -> f (R,nx) [, J (R,nx,nx+nu)]."""
        R = X.shape[0]
        eng = self._rows(R)
        if eng.n_extra:
            eng.bind_extra(E.reshape(R, 1, eng.n_extra))
        Z = torch.cat([torch.zeros_like(X), U], dim=1).contiguous()
        res = eng.eval(Z, X.contiguous(), want=("g", "jac_tiles") if want_jac else ("g",))
        f = res["g"]
        return (f, res["jac_tiles"].reshape(R, self.x_dim, self.x_dim + self.u_dim)) if want_jac else f

    def _to_dev(self, x, u, p=None, tvp=None):
        eng = self._rows(np.asarray(x).shape[0])
        ex = self.gather_extra(np.asarray(x).shape[0], p, tvp)
        return eng.to_device(x), eng.to_device(u), (None if ex is None else eng.to_device(ex))

    def forward(self, x, u, p=None, tvp=None):
        X, U, E = self._to_dev(x, u, p, tvp)
        return self.rows_batch(X, U, want_jac=False, E=E).to("cpu", torch.float64).numpy()

    def jacobian(self, x, u, p=None, tvp=None):
        """Block layout of the reference, (H*nx, H*nx + H*nu) with columns [all x | all u]
        (model/tensorflow.py:68-73); only the t == t' blocks are non-zero."""
        X, U, E = self._to_dev(x, u, p, tvp)
        _, J = self.rows_batch(X, U, E=E)
        J = J.to("cpu", torch.float64).numpy()
        H, nx, nu = J.shape[0], self.x_dim, self.u_dim
        out = np.zeros((H * nx, H * nx + H * nu))
        for t in range(H):
            out[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = J[t, :, :nx]
            out[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu] = J[t, :, nx:]
        return out

    def hessian(self, x, u, p=None, tvp=None):
        """(H, nx, H*(nx+nu), H*(nx+nu)) in the same block column order (model/tensorflow.py:87-109).
        One device call per output component (one-hot multipliers)."""
        X, U, E = self._to_dev(x, u, p, tvp)
        R, nx, nu = X.shape[0], self.x_dim, self.u_dim
        nin = nx + nu
        eng = self._rows(R)
        if eng.n_extra:
            eng.bind_extra(E.reshape(R, 1, eng.n_extra))
        Z = torch.cat([torch.zeros_like(X), U], dim=1).contiguous()
        sigma = torch.zeros(R, dtype=eng.dtype, device=eng.device)
        n = R * nin
        out = np.zeros((R, nx, n, n))
        for k in range(nx):
            lam = torch.zeros(R, eng.m, dtype=eng.dtype, device=eng.device)
            lam[:, k] = 1.0
            blk = eng.hess(Z, X.contiguous(), lam, sigma, want=("hblocks",))["hblocks"]
            blk = blk.reshape(R, nin, nin).to("cpu", torch.float64).numpy()
            for t in range(R):
                xs, us = slice(t * nx, (t + 1) * nx), slice(R * nx + t * nu, R * nx + (t + 1) * nu)
                out[t, k][xs, xs] = blk[t, :nx, :nx]
                out[t, k][xs, us] = blk[t, :nx, nx:]
                out[t, k][us, xs] = blk[t, nx:, :nx]
                out[t, k][us, us] = blk[t, nx:, nx:]
        return out
