#!/usr/bin/env python3
"""One batched solve at C2 / C3 dims (B=1024), deferred backtracking unless told otherwise: wall time, plus -- under
rocprofv3 --kernel-trace --stats -- the per-kernel table of exactly this solve.   python tools/solver_one.py [c2|c3] [max_iter] [linesearch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ls = sys.argv[3] if len(sys.argv) > 3 else "auto"
B = int(os.environ.get("NEMPC_TOOL_B", "1024"))
if cfg == "c2":
    nx, nu, H, hidden, integ, DT, dt = 2, 1, 20, [64, 64], "discret", 1.0, torch.float64
else:
    nx, nu, H, hidden, integ, DT, dt = 6, 3, 30, [128, 128, 128], "rk4", 0.1, torch.float32
net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B)
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
eng.solve(X0, lb=lb, ub=-lb, max_iter=5, linesearch=ls)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    Z, st, it = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi, linesearch=ls)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) * 1e3
    print(f"{cfg} B={B} max_iter={mi} ls={ls}: {dt_ms:.2f} ms, {it} iterations, {int((st == 0).sum())} converged")
