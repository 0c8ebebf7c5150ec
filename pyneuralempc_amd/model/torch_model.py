"""Dynamics given as a differentiable torch callable -- the counterpart of the reference's ``DiffDiscretJaxModel``
(model/jax.py:32-88), which wraps an arbitrary JAX function and differentiates it with jax.jacobian / jax.hessian.

An arbitrary Python function cannot be compiled into the HIP kernels (those evaluate dense networks given by their
weights: ``MLPModel`` / ``KerasTFModel``).  This class is the general-purpose plug-in for everything else -- analytic
dynamics, physics-informed hybrids, networks with layers the kernels do not have: the function is differentiated with
``torch.func`` on the device its tensors live on and the results are returned in the reference's layouts, so the
integrators' generic algebra (integrator/host.py) and the unfused solver glue take it unchanged.  One host round trip per
callback: this is the compatibility path, not the fast one (same split as ``TorchObjectifFunc`` / ``QuadraticObjective``
on the objective side).

    def f(x, u, p=None, tvp=None):          # x (H, x_dim), u (H, u_dim) torch tensors -> (H, x_dim)
        return torch.stack([x[:, 1], -torch.sin(x[:, 0]) + u[:, 0]], dim=1)
    model = TorchModel(f, x_dim=2, u_dim=1)                     # vector_mode=True: f maps the whole trajectory
    model = TorchModel(f_row, 2, 1, vector_mode=False)          # f_row(x (x_dim,), u (u_dim,), p, tvp (tvp_dim,)) per row
"""
import numpy as np
import torch

from .base import Model


class TorchModel(Model):
    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, vector_mode=True, safe_mode=True,
                 device="cuda", dtype=torch.float64):
        """forward_func(x, u, p=None, tvp=None): torch tensors in, torch tensor out, differentiable by torch.func.
        vector_mode=True (the reference's only implemented mode, model/jax.py:45-49): the function maps the whole
        (H, .) trajectory -- its Jacobian is taken densely, rows may couple; vector_mode=False: the function maps ONE row
        and is vmapped -- derivatives cost O(H) instead of O(H^2).  safe_mode: differentiate once at zeros in the
        constructor and refuse a function torch.func cannot differentiate (model/jax.py:23-39)."""
        super().__init__(x_dim, u_dim, int(p_dim or 0), int(tvp_dim or 0))
        self.forward_func = forward_func
        self.vector_mode = bool(vector_mode)
        self.device, self.dtype = torch.device(device), dtype
        if safe_mode:
            try:
                z = np.zeros
                self.jacobian(z((2, x_dim)), z((2, u_dim)), p=z(self.p_dim) if self.p_dim else None,
                              tvp=z((2, self.tvp_dim)) if self.tvp_dim else None)
            except Exception as e:      # noqa: BLE001
                raise ValueError(f"Your function is not differentiable w.r.t the torch.func library ({type(e).__name__}: {e})")

    def _t(self, a):
        return None if a is None else torch.as_tensor(np.asarray(a, dtype=np.float64)).to(self.device, self.dtype)

    @staticmethod
    def _np(t):
        return t.detach().to("cpu", torch.float64).numpy()

    def _row_fn(self, p, has_tvp):
        f = self.forward_func
        if has_tvp:
            return lambda xr, ur, tr: f(xr, ur, p=p, tvp=tr)
        return lambda xr, ur: f(xr, ur, p=p, tvp=None)

    # ---- (H, x_dim)
    def forward(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        with torch.no_grad():
            if self.vector_mode:
                return self._np(self.forward_func(X, U, p=P, tvp=T))
            args = (X, U) + ((T,) if T is not None else ())
            return self._np(torch.func.vmap(self._row_fn(P, T is not None))(*args))

    def _row_tiles(self, X, U, P, T, order):
        """per-row derivative blocks of a row function: order 1 -> (H, nx, nx+nu), order 2 -> (H, nx, nin, nin)"""
        fn = self._row_fn(P, T is not None)
        args = (X, U) + ((T,) if T is not None else ())
        in_dims = (0,) * len(args)
        if order == 1:
            jx, ju = torch.func.vmap(torch.func.jacrev(fn, argnums=(0, 1)), in_dims=in_dims)(*args)
            return torch.cat([jx, ju], dim=2)
        (hxx, hxu), (hux, huu) = torch.func.vmap(torch.func.hessian(fn, argnums=(0, 1)), in_dims=in_dims)(*args)
        return torch.cat([torch.cat([hxx, hxu], dim=3), torch.cat([hux, huu], dim=3)], dim=2)

    # ---- (H*x_dim, H*x_dim + H*u_dim), columns [all x | all u]  (model/jax.py:52-64, model/tensorflow.py:68-73)
    def jacobian(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        H, nx, nu = X.shape[0], self.x_dim, self.u_dim
        if self.vector_mode:
            jx, ju = torch.func.jacrev(lambda a, b: self.forward_func(a, b, p=P, tvp=T), argnums=(0, 1))(X, U)
            return self._np(torch.cat([jx.reshape(H * nx, H * nx), ju.reshape(H * nx, H * nu)], dim=1))
        tiles = self._np(self._row_tiles(X, U, P, T, 1))
        out = np.zeros((H * nx, H * (nx + nu)))
        for t in range(H):
            out[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = tiles[t, :, :nx]
            out[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu] = tiles[t, :, nx:]
        return out

    # ---- (H, x_dim, n, n), n = H*(x_dim + u_dim), same column order on both trailing axes  (model/jax.py:66-88)
    def hessian(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        H, nx, nu = X.shape[0], self.x_dim, self.u_dim
        if self.vector_mode:
            (hxx, hxu), (hux, huu) = torch.func.hessian(lambda a, b: self.forward_func(a, b, p=P, tvp=T),
                                                        argnums=(0, 1))(X, U)
            top = torch.cat([hxx.reshape(H, nx, H * nx, H * nx), hxu.reshape(H, nx, H * nx, H * nu)], dim=3)
            bot = torch.cat([hux.reshape(H, nx, H * nu, H * nx), huu.reshape(H, nx, H * nu, H * nu)], dim=3)
            return self._np(torch.cat([top, bot], dim=2))
        blk = self._np(self._row_tiles(X, U, P, T, 2))
        n = H * (nx + nu)
        out = np.zeros((H, nx, n, n))
        for t in range(H):
            xs, us = slice(t * nx, (t + 1) * nx), slice(H * nx + t * nu, H * nx + (t + 1) * nu)
            out[t][:, xs, xs] = blk[t, :, :nx, :nx]
            out[t][:, xs, us] = blk[t, :, :nx, nx:]
            out[t][:, us, xs] = blk[t, :, nx:, :nx]
            out[t][:, us, us] = blk[t, :, nx:, nx:]
        return out


class TorchModelRollingWindow(TorchModel):
    """Rolling-window dynamics given as a differentiable torch callable -- the counterpart of the reference's
    ``DiffDiscretJaxModelRollingWindow`` (model/jax.py:93-259): every row sees the last ``rolling_window`` states and
    controls,

        forward_func(x_slided (H, w * x_dim), u_slided (H, w * u_dim), p=None, tvp=tvp_slided (H, w * tvp_dim) | None)

    oldest row first inside a window (``_slide_input``, model/jax.py:141-153); ``set_prev_data`` supplies the w - 1 rows in
    front of the horizon (model/jax.py:119-129).  The reference differentiates with respect to the SLIDED inputs and
    projects the result back onto the decision variables with 0/1 matrices (``gen_jac_proj_mat``, model/jax.py:8-20,183-194,
    211-255); here the window gather is part of the differentiated function -- a composition of torch indexing ops -- so
    ``torch.func`` returns the derivatives with respect to the decision variables directly: the same numbers (the chain
    rule through a 0/1 selection), the same layouts, no projection matrices.  Like the reference, only forward_rolling=True
    and vector_mode=True exist.  A rolling-window NETWORK given by its weights runs in the HIP kernels instead
    (``MLPModelRollingInput`` / ``KerasTFModelRollingInput``)."""

    def __init__(self, forward_func, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, rolling_window=1, forward_rolling=True,
                 vector_mode=True, safe_mode=True, device="cuda", dtype=torch.float64):
        if not forward_rolling:
            raise NotImplementedError("Sorry ='(")                       # (model/jax.py:108-109)
        if not vector_mode:
            raise NotImplementedError("")                               # (model/jax.py:161-162)
        if not isinstance(rolling_window, int) or rolling_window < 1:
            raise ValueError("Your rolling windows need to be an integer gretter than 1.")
        self.rolling_window = rolling_window
        self.forward_rolling = forward_rolling
        self.prev_x = self.prev_u = self.prev_tvp = None
        Model.__init__(self, x_dim, u_dim, int(p_dim or 0), int(tvp_dim or 0))
        self.forward_func = forward_func
        self.vector_mode = True
        self.device, self.dtype = torch.device(device), dtype
        if safe_mode:
            keep = (self.prev_x, self.prev_u, self.prev_tvp)
            try:
                w = rolling_window
                self.set_prev_data(np.zeros((w - 1, x_dim)), np.zeros((w - 1, u_dim)),
                                   np.zeros((w - 1, self.tvp_dim)) if self.tvp_dim else None)
                self.jacobian(np.zeros((2, x_dim)), np.zeros((2, u_dim)), p=np.zeros(self.p_dim) if self.p_dim else None,
                              tvp=np.zeros((2, self.tvp_dim)) if self.tvp_dim else None)
            except Exception as e:      # noqa: BLE001
                raise ValueError(f"Your function is not differentiable w.r.t the torch.func library ({type(e).__name__}: {e})")
            finally:
                self.prev_x, self.prev_u, self.prev_tvp = keep

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

    def _windowed(self, P, T):
        """the function of the DECISION rows (X (H, x_dim), U (H, u_dim)) that torch.func differentiates"""
        assert self.prev_x is not None and self.prev_u is not None, \
            "You must give history window with set_prev_data before calling any inferance function."
        w = self.rolling_window
        px, pu = self._t(self.prev_x), self._t(self.prev_u)
        tv = None
        if T is not None:
            tv = self._slide(torch.cat([self._t(self.prev_tvp), T], dim=0), w)

        def g(X, U):
            return self.forward_func(self._slide(torch.cat([px, X], dim=0), w), self._slide(torch.cat([pu, U], dim=0), w),
                                     p=P, tvp=tv)
        return g

    @staticmethod
    def _slide(a, w):
        n = a.shape[0] - w + 1
        return torch.stack([a[i:i + w].reshape(-1) for i in range(n)])

    def forward(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        with torch.no_grad():
            return self._np(self._windowed(P, T)(X, U))

    def jacobian(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        H, nx, nu = X.shape[0], self.x_dim, self.u_dim
        jx, ju = torch.func.jacrev(self._windowed(P, T), argnums=(0, 1))(X, U)
        return self._np(torch.cat([jx.reshape(H * nx, H * nx), ju.reshape(H * nx, H * nu)], dim=1))

    def hessian(self, x, u, p=None, tvp=None):
        X, U, P, T = self._t(x), self._t(u), self._t(p), self._t(tvp)
        H, nx, nu = X.shape[0], self.x_dim, self.u_dim
        (hxx, hxu), (hux, huu) = torch.func.hessian(self._windowed(P, T), argnums=(0, 1))(X, U)
        top = torch.cat([hxx.reshape(H, nx, H * nx, H * nx), hxu.reshape(H, nx, H * nx, H * nu)], dim=3)
        bot = torch.cat([hux.reshape(H, nx, H * nu, H * nx), huu.reshape(H, nx, H * nu, H * nu)], dim=3)
        return self._np(torch.cat([top, bot], dim=2))
