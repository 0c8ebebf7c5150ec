"""Rolling-window network models on the device (reference: KerasTFModelRollingInput model/tensorflow.py:132-340,
DiffDiscretJaxModelRollingWindow model/jax.py:93-259).

The network of step t reads the last ``rolling_window`` states and controls
    xi_t = [ x_ext[t : t+w].ravel() | u_ext[t : t+w].ravel() | tvp window | p ],   x_ext = [prev_x ; x],  u_ext = [prev_u ; u]
(oldest first, or newest first with ``forward_rolling=False``), the history ``prev_x / prev_u`` being supplied with
``set_prev_data`` before each solve.  The reference differentiates through that gather with TF / JAX and projection
matrices; here the gather is part of the row kernels' input staging and the band structure of the Jacobian / Hessian is
built once on the host (csrc/nempc_api.hip window_var)."""
import numpy as np
import torch

from .mlp import MLPModel
from ..engine import CallbackEngine


class MLPModelRollingInput(MLPModel):
    def __init__(self, weights, biases, x_dim, u_dim, p_dim=0, tvp_dim=0, rolling_window=2, forward_rolling=True,
                 dtype=torch.float64, device="cuda", kernel="auto", activations=None):
        if not isinstance(rolling_window, int) or rolling_window < 1:
            raise ValueError("Your rolling windows need to be an integer gretter than 1.")
        p_dim, tvp_dim = int(p_dim or 0), int(tvp_dim or 0)
        w0 = np.asarray(weights[0])
        if w0.shape[0] != rolling_window * (x_dim + u_dim + tvp_dim) + p_dim:
            raise ValueError("Your model do not provide a suitable input dim ! \n It must get rolling_window * "
                             "(x_dim + u_dim + tvp_dim) + p_dim inputs.")
        self.rolling_window = rolling_window
        self.forward_rolling = bool(forward_rolling)
        self.prev_x, self.prev_u, self.prev_tvp = None, None, None
        # MLPModel checks in = x + u + p + tvp: present the windowed widths to it
        MLPModel.__init__(self, weights, biases, x_dim, u_dim, p_dim, tvp_dim, dtype=dtype, device=device,
                          kernel=kernel, _input_width=w0.shape[0], activations=activations)

    def __getstate__(self):   # tensorflow.py:168-175: the history does not travel
        d = MLPModel.__getstate__(self)
        d["prev_x"] = d["prev_u"] = d["prev_tvp"] = None
        return d

    def set_prev_data(self, x_prev, u_prev, tvp_prev=None):
        w = self.rolling_window
        x_prev, u_prev = np.asarray(x_prev, dtype=np.float64), np.asarray(u_prev, dtype=np.float64)
        assert x_prev.shape == (w - 1, self.x_dim), \
            f"Your x prev tensor must have the following shape {(w - 1, self.x_dim)} (received : {x_prev.shape})"
        assert u_prev.shape == (w - 1, self.u_dim), \
            f"Your u prev tensor must have the following shape {(w - 1, self.u_dim)} (received : {u_prev.shape})"
        self.prev_x, self.prev_u = x_prev, u_prev
        if tvp_prev is not None:
            tvp_prev = np.asarray(tvp_prev, dtype=np.float64)
            assert tvp_prev.shape == (w - 1, self.tvp_dim), \
                f"Your tvp prev tensor must have the following shape {(w - 1, self.tvp_dim)} (received : {tvp_prev.shape})"
            self.prev_tvp = tvp_prev

    # -- engine plumbing ------------------------------------------------------------------------
    @property
    def n_extra(self):
        return self.rolling_window * self.tvp_dim + self.p_dim

    def make_engine(self, H, integrator, DT=1.0, max_batch=1):
        return CallbackEngine(self.weights, self.biases, H, self.x_dim, self.u_dim, integrator=integrator, DT=DT,
                              dtype=self.dtype, device=self.device, max_batch=max_batch, kernel=self.kernel,
                              n_extra=self.n_extra, rolling_window=self.rolling_window,
                              forward_rolling=self.forward_rolling, activations=self.activations)

    def _roll(self, prev, cur):
        ext = np.concatenate([prev, cur], axis=0)
        w = self.rolling_window
        order = slice(None) if self.forward_rolling else slice(None, None, -1)
        return np.stack([ext[t:t + w][order].reshape(-1) for t in range(cur.shape[0])], axis=0)

    def gather_extra(self, rows, p=None, tvp=None):
        """(rows, w*tvp_dim + p_dim): the time-varying parameters rolled like the states, then p
        (_gather_input_V2, model/tensorflow.py:218-233)."""
        if self.n_extra == 0:
            return None
        parts = []
        if self.tvp_dim:
            if tvp is None:
                raise ValueError("this model has tvp_dim > 0: pass tvp (H, tvp_dim)")
            assert self.prev_tvp is not None or self.rolling_window == 1, \
                "You must give history window with set_prev_data before calling any inferance function."
            tvp = np.asarray(tvp, dtype=np.float64)
            assert tvp.shape == (rows, self.tvp_dim), "tvp first dim must set according to the horizon size !"
            prev = self.prev_tvp if self.rolling_window > 1 else np.zeros((0, self.tvp_dim))
            parts.append(self._roll(prev, tvp))
        if self.p_dim:
            if p is None:
                raise ValueError("this model has p_dim > 0: pass p (p_dim,)")
            parts.append(np.tile(np.asarray(p, dtype=np.float64).reshape(1, self.p_dim), (rows, 1)))
        return np.concatenate(parts, axis=1)

    def bind_inputs(self, eng, p=None, tvp=None):
        bound = MLPModel.bind_inputs(self, eng, p, tvp)
        if self.rolling_window > 1:
            assert (self.prev_x is not None) and (self.prev_u is not None), \
                "You must give history window with set_prev_data before calling any inferance function."
            eng.bind_history(eng.to_device(self.prev_x[None]), eng.to_device(self.prev_u[None]))
            bound = True
        return bound

    # -- reference Model signatures (one trajectory, NumPy in / out) ----------------------------------
    def _slots(self, H):
        """For every step t and window slot j: the row of x (resp. u) it reads, negative = history."""
        w = self.rolling_window
        off = np.arange(w) - (w - 1) if self.forward_rolling else -np.arange(w)
        return np.arange(H)[:, None] + off[None, :]

    def _trajectory_engine(self, x, u, p, tvp):
        # a UNITY problem over the whole trajectory: x0 = x[0], states = [x[1:] ; 0] so that [x0 ; states[:-1]] == x
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        H = x.shape[0]
        key = ("traj", H)
        if self._row_engine is None or self._row_engine[0] != key:
            self._row_engine = (key, self.make_engine(H, "unity", max_batch=1))
        eng = self._row_engine[1]
        self.bind_inputs(eng, p, tvp)
        states = np.concatenate([x[1:], np.zeros((1, self.x_dim))], axis=0)
        z = np.concatenate([states.reshape(-1), u.reshape(-1)])
        return eng, eng.to_device(z[None]), eng.to_device(x[:1]), states

    def forward(self, x, u, p=None, tvp=None):
        eng, Z, X0, states = self._trajectory_engine(x, u, p, tvp)
        g = eng.eval(Z, X0, want=("g",))["g"][0].to("cpu", torch.float64).numpy()
        return g[:states.size].reshape(states.shape) + states      # UNITY defect is f - x_t

    def _columns(self, H):
        """tile column -> column of the [all x | all u] layout for every step (H, w*(nx+nu)), -1 = history."""
        nx, nu, w = self.x_dim, self.u_dim, self.rolling_window
        tau = self._slots(H)                                            # (H, w)
        cx = np.where(tau[:, :, None] >= 0, tau[:, :, None] * nx + np.arange(nx)[None, None, :], -1)
        cu = np.where(tau[:, :, None] >= 0, H * nx + tau[:, :, None] * nu + np.arange(nu)[None, None, :], -1)
        return np.concatenate([cx.reshape(H, w * nx), cu.reshape(H, w * nu)], axis=1)

    def jacobian(self, x, u, p=None, tvp=None):
        """(H*nx, H*nx + H*nu), columns [all x | all u] like model/tensorflow.py:253-262: lower block-banded."""
        eng, Z, X0, _ = self._trajectory_engine(x, u, p, tvp)
        tiles = eng.eval(Z, X0, want=("jac_tiles",))["jac_tiles"][0].to("cpu", torch.float64).numpy()
        H, nx, nu = tiles.shape[0], self.x_dim, self.u_dim
        cols = self._columns(H)
        out = np.zeros((H * nx, H * (nx + nu)))
        for t in range(H):
            keep = cols[t] >= 0
            out[t * nx:(t + 1) * nx, cols[t][keep]] = tiles[t][:, keep]
        return out

    def hessian(self, x, u, p=None, tvp=None):
        """(H, nx, n, n) in the same column order (model/tensorflow.py:298-340); one device call per output."""
        eng, Z, X0, _ = self._trajectory_engine(x, u, p, tvp)
        H, nx, nu = eng.H, self.x_dim, self.u_dim
        n = H * (nx + nu)
        cols = self._columns(H)
        sigma = torch.zeros(1, dtype=eng.dtype, device=eng.device)
        out = np.zeros((H, nx, n, n))
        for k in range(nx):
            lam = torch.zeros(1, eng.m, dtype=eng.dtype, device=eng.device)
            lam[0, k::nx] = 1.0
            blk = eng.hess(Z, X0, lam, sigma, want=("hblocks",))["hblocks"][0].to("cpu", torch.float64).numpy()
            for t in range(H):
                keep = np.nonzero(cols[t] >= 0)[0]
                out[t, k][np.ix_(cols[t][keep], cols[t][keep])] = blk[t][np.ix_(keep, keep)]
        return out
