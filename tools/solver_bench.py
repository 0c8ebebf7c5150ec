#!/usr/bin/env python3
"""Batched-solver timing (C2 dims): solved MPC problems per second, convergence counts."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nx, nu, H = 2, 1, 20
net = orc.MLP.random(3, [64, 64], 2, seed=0); net.W[-1] *= 0.2; net.b[-1] *= 0.2
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(11).uniform(-1, 1, size=(B, nx)))
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)]); ub = -lb
for bounded in (True, False):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        Z, st, it = eng.solve(X0, lb=lb if bounded else None, ub=ub if bounded else None, max_iter=mi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{'bounded' if bounded else 'unbounded'} B={B} max_iter={mi}: {dt*1e3:.2f} ms, {it} iterations, "
          f"{int((st == 0).sum())}/{B} converged -> {B/dt:.0f} MPC solves/s ({dt/it*1e6:.0f} us per iteration)")
