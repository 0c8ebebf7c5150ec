"""Integrator algebra over ANY ``Model`` plug-in, on the host (reference: integrator/discret.py:13-81,
unity.py:15-81, rk4.py:57-285).

The device integrators evaluate dense networks inside the HIP kernels.  A model that is not such a network --
``model.TorchModel`` (an arbitrary differentiable callable, the reference's ``DiffDiscretJaxModel``), or a user's own
``Model`` subclass -- has no kernel; for those the Discret / Unity / RK4 classes fall back to this module, which does
what the reference's integrators do with a model's outputs: call ``model.forward / jacobian / hessian`` (block layout,
columns ``[all x_{t-1} | all u]``) and place the results into the defect vector, its dense Jacobian ``(H nx, n)`` and
its Hessian ``(H nx, n, n)`` over ``z = [states | controls]``.

Formulation (one index map instead of the reference's slice arithmetic): the model is evaluated on
``x_prev = [x0 ; states[:-1]]``, so model input column ``j`` of the x-part belongs to decision variable ``j - nx`` (the
first block, ``x0``, is data and has no column); control columns map to themselves.  ``var_of_col`` holds that map.
"""
import numpy as np


def _x_prev(x, x0):
    return np.concatenate([np.asarray(x0, dtype=np.float64).reshape(1, -1), np.asarray(x, dtype=np.float64)[:-1]], axis=0)


def var_of_col(H, nx, nu):
    """decision-variable index of every model input column ([all x_prev | all u]); -1 for the x0 block"""
    m = np.arange(H * (nx + nu)) - nx
    m[:nx] = -1
    m[H * nx:] = np.arange(H * nx, H * (nx + nu))
    return m


def _row_tiles(model_jac, H, nx, nu):
    """(H, nx, nx+nu): the t == t' blocks of a block-layout Jacobian (rk4.py:85-92)"""
    out = np.empty((H, nx, nx + nu))
    for t in range(H):
        out[t, :, :nx] = model_jac[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx]
        out[t, :, nx:] = model_jac[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu]
    return out


def _row_hessians(model_hess, H, nx, nu):
    """(H, nx, nin, nin): the t == t' blocks of a block-layout Hessian (rk4.py:94-110)"""
    nin = nx + nu
    out = np.empty((H, nx, nin, nin))
    for t in range(H):
        idx = np.concatenate([np.arange(t * nx, (t + 1) * nx), H * nx + np.arange(t * nu, (t + 1) * nu)])
        out[t] = model_hess[t][:, idx][:, :, idx]
    return out


def _scatter_tiles(tiles, H, nx, nu, identity):
    """dense (H nx, n) Jacobian of the defects Phi(x_{t-1}, u_t) - x_t from per-row tiles dPhi/d[x_{t-1} | u_t]"""
    n = H * (nx + nu)
    J = np.zeros((H * nx, n))
    J[:, :H * nx] = -np.eye(H * nx)
    for t in range(H):
        r = slice(t * nx, (t + 1) * nx)
        if t > 0:
            J[r, (t - 1) * nx:t * nx] += tiles[t, :, :nx] + (np.eye(nx) if identity else 0.0)
        J[r, H * nx + t * nu:H * nx + (t + 1) * nu] += tiles[t, :, nx:]
    return J


class HostAlgebra:
    """forward / jacobian / hessian of one integrator kind over a generic Model."""

    def __init__(self, model, H, kind, DT=1.0):
        self.model, self.H, self.kind, self.DT = model, int(H), kind, float(DT)
        self.nx, self.nu = model.x_dim, model.u_dim

    # ---- defects (H nx,)
    def forward(self, x, u, x0, p=None, tvp=None):
        assert len(x.shape) == 2 and len(u.shape) == 2, "x and u tensor must have dim 2"
        assert len(x0.shape) == 1, "x0 shape must have dim 1"
        x = np.asarray(x, dtype=np.float64)
        xp = _x_prev(x, x0)
        f = lambda xs: np.asarray(self.model.forward(xs, u, p=p, tvp=tvp), dtype=np.float64)   # noqa: E731
        if self.kind == "unity":
            phi = f(xp)
        elif self.kind == "discret":
            phi = xp + f(xp)
        else:
            DT = self.DT
            k1 = f(xp); k2 = f(xp + 0.5 * DT * k1); k3 = f(xp + 0.5 * DT * k2); k4 = f(xp + DT * k3)
            phi = xp + DT / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        return (phi - x).reshape(-1)

    # ---- RK4 stage sweep on per-row tiles: stage inputs, Jacobians J_s, chain Jacobians dk_s = J_s R_s,
    #      R_s = I + c_s DT [dk_{s-1} ; 0]   (rk4.py:137-159)
    def _rk4_stages(self, xp, u, p, tvp, want_hess):
        H, nx, nu, DT = self.H, self.nx, self.nu, self.DT
        nin = nx + nu
        cs = (0.0, 0.5 * DT, 0.5 * DT, DT)
        k = np.zeros((H, nx))
        dk = np.zeros((H, nx, nin))
        rec = []
        for s in range(4):
            xs = xp + cs[s] * k
            R = np.broadcast_to(np.eye(nin), (H, nin, nin)).copy()
            R[:, :nx, :] += cs[s] * dk
            J = _row_tiles(np.asarray(self.model.jacobian(xs, u, p=p, tvp=tvp)), H, nx, nu)
            Hs = _row_hessians(np.asarray(self.model.hessian(xs, u, p=p, tvp=tvp)), H, nx, nu) if want_hess else None
            k = np.asarray(self.model.forward(xs, u, p=p, tvp=tvp), dtype=np.float64)
            dk = J @ R
            rec.append((J, R, dk, Hs))
        return rec

    # ---- dense Jacobian (H nx, n)
    def jacobian(self, x, u, x0, p=None, tvp=None):
        H, nx, nu = self.H, self.nx, self.nu
        xp = _x_prev(x, x0)
        if self.kind == "rk4":
            w = (1.0, 2.0, 2.0, 1.0)
            tiles = self.DT / 6.0 * sum(wi * r[2] for wi, r in zip(w, self._rk4_stages(xp, u, p, tvp, False)))
            return _scatter_tiles(tiles, H, nx, nu, identity=True)
        mj = np.asarray(self.model.jacobian(xp, u, p=p, tvp=tvp), dtype=np.float64)
        vc = var_of_col(H, nx, nu)
        keep = vc >= 0
        J = np.zeros((H * nx, H * (nx + nu)))
        J[:, :H * nx] = -np.eye(H * nx)
        J[:, vc[keep]] += mj[:, keep]
        if self.kind == "discret":                          # d x_{t-1} / d x_{t-1}: the sub-diagonal identity
            J[nx:, :(H - 1) * nx] += np.eye((H - 1) * nx)
        return J

    # ---- Hessian of every defect row (H nx, n, n)
    def hessian(self, x, u, x0, p=None, tvp=None):
        H, nx, nu = self.H, self.nx, self.nu
        n = H * (nx + nu)
        xp = _x_prev(x, x0)
        out = np.zeros((H * nx, n, n))
        if self.kind != "rk4":
            mh = np.asarray(self.model.hessian(xp, u, p=p, tvp=tvp), dtype=np.float64).reshape(H * nx, n, n)
            vc = var_of_col(H, nx, nu)
            keep = np.nonzero(vc >= 0)[0]
            out[np.ix_(np.arange(H * nx), vc[keep], vc[keep])] = mh[np.ix_(np.arange(H * nx), keep, keep)]
            return out
        # RK4, any dims (the reference hard-wires nx + nu = 3, rk4.py:246-261): second-order chain rule per row
        #   h_s = R_s^T Hf(xi_s) R_s + c_s DT sum_j J_s[:, j] h_{s-1}[j],    d2 Phi = DT/6 (h_1 + 2 h_2 + 2 h_3 + h_4)
        cs = (0.0, 0.5 * self.DT, 0.5 * self.DT, self.DT)
        w = (1.0, 2.0, 2.0, 1.0)
        nin = nx + nu
        h = np.zeros((H, nx, nin, nin))
        acc = np.zeros_like(h)
        for s, (J, R, _, Hs) in enumerate(self._rk4_stages(xp, u, p, tvp, True)):
            h = np.einsum("tap,tkab,tbq->tkpq", R, Hs, R) + cs[s] * np.einsum("tkj,tjpq->tkpq", J[:, :, :nx], h)
            acc += w[s] * h
        acc *= self.DT / 6.0
        for t in range(H):
            cols = np.concatenate([np.arange((t - 1) * nx, t * nx) if t > 0 else -np.ones(nx, dtype=np.int64),
                                   H * nx + np.arange(t * nu, (t + 1) * nu)])
            keep = np.nonzero(cols >= 0)[0]
            out[np.ix_(np.arange(t * nx, (t + 1) * nx), cols[keep], cols[keep])] = acc[t][np.ix_(np.arange(nx), keep, keep)]
        return out
