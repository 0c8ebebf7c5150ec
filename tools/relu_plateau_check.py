#!/usr/bin/env python3
"""Are the points where the batched solver stalls on a relu network local minima?  For every problem of the relu test batch
(tests/test_gpu_solver.py::test_batched_solve_with_other_activations) that did not meet the step test: the objective in
the reduced space (controls only, states by roll-out, which keeps the defects at zero) under random perturbations of the
controls, and under a derivative-free descent (Nelder-Mead) started from the solver's point.
   python tools/relu_plateau_check.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
from scipy.optimize import minimize
nx, nu, H, B = 2, 1, 12, 32
net = orc.MLP.random(nx + nu, [64, 64], nx, seed=3, activations="relu")
net.W[-1] *= 0.2; net.b[-1] *= 0.2
Q, Rw = np.eye(nx), 0.1 * np.eye(nu)
lb = np.concatenate([np.full(H * nx, -10.0), np.full(H * nu, -0.5)])
X0 = np.random.default_rng(5).uniform(-0.6, 0.6, size=(B, nx))
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, activations="relu")
eng.set_objective(Q=Q, R=Rw)
Z, status, iters = eng.solve(eng.to_device(X0), lb=lb, ub=-lb, max_iter=int(os.environ.get("ITERS", "300")))
Z, status = Z.cpu().numpy(), status.cpu().numpy()
print("status counts", {int(s): int((status == s).sum()) for s in np.unique(status)})

def rollout_cost(u, x0):
    x, J = x0.copy(), 0.0
    for t in range(H):
        x = x + net.forward(np.concatenate([x, u[t * nu:(t + 1) * nu]])[None])[0]
        J += x @ Q @ x + u[t * nu:(t + 1) * nu] @ Rw @ u[t * nu:(t + 1) * nu]
    return J

rng = np.random.default_rng(0)
for i in range(B):
    u = Z[i, H * nx:].copy()
    xs = Z[i, :H * nx]
    J0 = rollout_cost(u, X0[i])
    # objective of the solver's own point (states as the solver left them) vs the roll-out: the defect it carries
    Jz = sum(xs[t*nx:(t+1)*nx] @ Q @ xs[t*nx:(t+1)*nx] for t in range(H)) + u @ u * 0.1
    best = 0.0
    for scale in (1e-2, 1e-3, 1e-4):
        for _ in range(300):
            d = rng.normal(size=u.size); d *= scale / np.linalg.norm(d)
            un = np.clip(u + d, -0.5, 0.5)
            best = min(best, rollout_cost(un, X0[i]) - J0)
    r = minimize(lambda v: rollout_cost(np.clip(v, -0.5, 0.5), X0[i]), u, method="Nelder-Mead",
                 options=dict(xatol=1e-9, fatol=1e-13, maxiter=4000, maxfev=8000))
    print(f"problem {i:2d} status {int(status[i]):2d}  J {J0:.9f} (solver's states: {Jz:.9f})  best random decrease {best:.2e}  "
          f"Nelder-Mead decrease {r.fun - J0:.2e}  |du| {np.abs(np.clip(r.x, -0.5, 0.5) - u).max():.1e}")
