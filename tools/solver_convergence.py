#!/usr/bin/env python3
"""Convergence profile of the batched solver: converged fraction vs iteration budget (C2 dims, bounded / unbounded)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B = 1024
nx, nu, H = 2, 1, 20
net = orc.MLP.random(3, [64, 64], 2, seed=0); net.W[-1] *= 0.2; net.b[-1] *= 0.2
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(11).uniform(-1, 1, size=(B, nx)))
for name, lbv in (("|x|<=3,|u|<=0.5", (3.0, 0.5)), ("|x|<=1.2,|u|<=0.3", (1.2, 0.3)), ("unbounded", None)):
    lb = None if lbv is None else np.concatenate([np.full(H * nx, -lbv[0]), np.full(H * nu, -lbv[1])])
    for mi in (20, 40, 80, 160, 320):
        torch.cuda.synchronize(); t = time.perf_counter()
        Z, st, it = eng.solve(X0, lb=lb, ub=None if lb is None else -lb, max_iter=mi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{name:20s} max_iter={mi:4d}: {int((st == 0).sum()):5d}/{B} converged in {it:4d} iterations, {dt*1e3:7.1f} ms")
