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
