#!/usr/bin/env python3
"""Batched solver: histogram of per-problem convergence iterations and wall time vs iteration budget, with and without
compaction.   python tools/solver_profile.py [c2|c3] [B]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if cfgname == "c2":
    nx, nu, H, hidden, kind, DT, dt = 2, 1, 20, [64, 64], "discret", 1.0, torch.float64
else:
    nx, nu, H, hidden, kind, DT, dt = 6, 3, 30, [128, 128, 128], "rk4", 0.1, torch.float32
net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=kind, DT=DT, dtype=dt, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
modes = [("primal-dual", True, "auto"), ("primal-dual", True, "deferred"), ("primal-dual", True, "loop"), ("primal", True, "loop")]
for mi in (40, 60, 80, 160):
    for barrier, compact, lsm in modes:
        eng.solve(X0, lb=lb, ub=-lb, max_iter=3)
        torch.cuda.synchronize(); t = time.perf_counter()
        Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi, compact=compact, return_iterations=True, barrier=barrier,
                                   linesearch=lsm)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t
        ok = (st == 0)
        p = per[ok].cpu().numpy()
        q = np.percentile(p, [50, 90, 95, 99]) if len(p) else [0] * 4
        print(f"{cfgname} B={B} max_iter={mi:3d} {barrier:11s} ls={lsm:8s} compact={int(compact)}: {dtm*1e3:7.2f} ms, {it:3d} iterations, {int(ok.sum()):5d}/{B} "
              f"converged ({int(ok.sum())/dtm:9.0f} solved/s); iterations to converge p50/p90/p95/p99 = "
              f"{q[0]:.0f}/{q[1]:.0f}/{q[2]:.0f}/{q[3]:.0f}")
