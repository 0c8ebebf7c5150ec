#!/usr/bin/env python3
"""Closed loop at C2 dims: B plants (the network itself is the plant), every MPC step solved on the device from the
previous solution shifted by one stage (warm start, small initial barrier parameter).  Reports iterations and wall time
per MPC step -- the serving figure behind `mpc_solved_per_s`, which is a cold-start number.
    python tools/closed_loop_bench.py [B] [steps] [max_iter]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 40
nx, nu, H = 2, 1, 20
net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="discret", DT=1.0, dtype=torch.float64, device="cuda:0", max_batch=B)
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
X = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
W = [torch.as_tensor(w, dtype=torch.float64, device="cuda:0") for w in net.W]
bb = [torch.as_tensor(b, dtype=torch.float64, device="cuda:0") for b in net.b]


def plant(x, u):       # Discret integrator: x+ = net(x, u)   (integrator/discret.py)
    h = torch.cat([x, u], dim=1)
    for k in range(len(W) - 1):
        h = torch.tanh(h @ W[k].T + bb[k]) if W[k].shape[1] == h.shape[1] else torch.tanh(h @ W[k] + bb[k])
    return h @ W[-1].T + bb[-1] if W[-1].shape[1] == h.shape[1] else h @ W[-1] + bb[-1]


eng.solve(X, lb=lb, ub=-lb, max_iter=3)
Zi = None
for k in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if Zi is None:
        Z, st, it, per = eng.solve(X, lb=lb, ub=-lb, max_iter=mi, return_iterations=True)
    else:
        Z, st, it, per = eng.solve(X, Zi, lb=lb, ub=-lb, max_iter=mi, return_iterations=True, mu_init=float(os.environ.get("MU0", "1e-4")))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = st == 0
    p = per[ok].float()
    print(f"step {k}: {dt*1e3:6.2f} ms, {it:3d} iterations, {int(ok.sum()):5d}/{B} solved ({int(ok.sum())/dt:9.0f} solved/s), "
          f"iterations to converge median {p.median().item() if len(p) else 0:.0f} max {p.max().item() if len(p) else 0:.0f}")
    xs, us = Z[:, :H * nx].reshape(B, H, nx), Z[:, H * nx:].reshape(B, H, nu)
    X = plant(X, us[:, 0, :])
    # shift by one stage, repeat the last; keep strictly inside the bounds
    xs2 = torch.cat([xs[:, 1:], xs[:, -1:]], dim=1).clamp(-2.999, 2.999)
    us2 = torch.cat([us[:, 1:], us[:, -1:]], dim=1).clamp(-0.499, 0.499)
    Zi = torch.cat([xs2.reshape(B, -1), us2.reshape(B, -1)], dim=1).contiguous()
