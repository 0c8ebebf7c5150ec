#!/usr/bin/env python3
"""Iterations (and wall time) at which 99 % of the batch has converged: C2 dims, B=1024, the bench's problem set.
python tools/solver_p99.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
cfg = bench.CONFIGS["c2"]
B, H, nx, nu = 1024, cfg["H"], cfg["nx"], cfg["nu"]
net = orc.MLP.random(nx + nu, cfg["hidden"], nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
Z, st, it, its = eng.solve(X0, lb=lb, ub=-lb, max_iter=400, return_iterations=True)
its = np.sort(its.cpu().numpy()[(st == 0).cpu().numpy()])
print("converged", len(its), "of", B, "after", it)
for frac in (0.95, 0.98, 0.99, 0.995):
    k = int(np.ceil(frac * B))
    if k <= len(its):
        n_it = int(its[k - 1])
        eng.solve(X0, lb=lb, ub=-lb, max_iter=n_it)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            Z2, st2, it2 = eng.solve(X0, lb=lb, ub=-lb, max_iter=n_it)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"{frac*100:.1f} % ({k} problems) converged within {n_it} iterations: {best:.2f} ms ({int((st2 == 0).sum())} converged in that run)")
