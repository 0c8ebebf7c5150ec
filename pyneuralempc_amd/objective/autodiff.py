"""User-supplied differentiable objective (reference: JAXObjectifFunc, objective/jax.py:16-90).

The reference differentiates an arbitrary JAX callable ``func(states, u, p, tvp) -> scalar`` with jax.grad /
jax.hessian.  JAX is not part of this build; the same contract is offered for a torch callable, differentiated with
torch.autograd on the device the tensors live on.  This is the general-purpose plug-in: it goes through the unfused
solver glue (one host round trip per callback).  The two objectives the reference's scripts use -- sum(u * c)
(examples/lotka_volterra/run.py:84-93) and sum((u - 2)^2) (test.py:55-60) -- are members of QuadraticObjective,
which is evaluated inside the fused HIP callback instead."""
import numpy as np
import torch

from .base import ObjectiveFunc


class TorchObjectifFunc(ObjectiveFunc):
    def __init__(self, func, device="cuda", dtype=torch.float64):
        super().__init__()
        self.func = func
        self.device, self.dtype = torch.device(device), dtype
        self.cached_hessian_structure = dict()

    def _t(self, a, grad=False):
        if a is None:
            return None
        t = torch.as_tensor(np.asarray(a, dtype=np.float64)).to(self.device, self.dtype)
        return t.requires_grad_(True) if grad else t

    def forward(self, states, u, p=None, tvp=None):
        with torch.no_grad():
            return float(self.func(self._t(states), self._t(u), self._t(p), self._t(tvp)))

    def gradient(self, states, u, p=None, tvp=None):
        s, c = self._t(states, True), self._t(u, True)
        val = self.func(s, c, self._t(p), self._t(tvp))
        gs, gu = torch.autograd.grad(val, (s, c), allow_unused=True)
        gs = torch.zeros_like(s) if gs is None else gs
        gu = torch.zeros_like(c) if gu is None else gu
        res = torch.cat([gs.reshape(-1), gu.reshape(-1)]).to("cpu", torch.float64).numpy()
        return np.nan_to_num(res, nan=0.0)          # objective/jax.py:40

    def hessian(self, states, u, p=None, tvp=None):
        """(n, n) over [states.ravel() | u.ravel()], assembled from the four blocks like objective/jax.py:43-57."""
        s, c = self._t(states), self._t(u)
        pp, tt = self._t(p), self._t(tvp)
        ns, nu = s.numel(), c.numel()
        blocks = torch.autograd.functional.hessian(lambda a, b: self.func(a, b, pp, tt), (s, c))
        top = torch.cat([blocks[0][0].reshape(ns, ns), blocks[0][1].reshape(ns, nu)], dim=1)
        bot = torch.cat([blocks[1][0].reshape(nu, ns), blocks[1][1].reshape(nu, nu)], dim=1)
        return torch.cat([top, bot], dim=0).to("cpu", torch.float64).numpy()

    def hessianstructure(self, H, model):
        key = (H, model)
        if key not in self.cached_hessian_structure:
            self.cached_hessian_structure[key] = self._compute_hessianstructure(H, model)
        return self.cached_hessian_structure[key]

    def _compute_hessianstructure(self, H, model, nb_sample=3):
        """OR of the non-zero patterns at three random points (objective/jax.py:67-90)."""
        hessian_map = None
        for _ in range(nb_sample):
            x_random = np.random.uniform(size=(H, model.x_dim))
            u_random = np.random.uniform(size=(H, model.u_dim))
            p_random = np.random.uniform(size=model.p_dim) if model.p_dim > 0 else None
            tvp_random = np.random.uniform(size=(H, model.tvp_dim)) if model.tvp_dim > 0 else None
            pattern = self.hessian(x_random, u_random, p=p_random, tvp=tvp_random) != 0.0
            hessian_map = pattern if hessian_map is None else (hessian_map | pattern)
        return hessian_map.astype(np.float64)
