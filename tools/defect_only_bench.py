#!/usr/bin/env python3
"""Defect-only (forward) launches vs full row launches: microseconds per launch at C2 / C3 / C5 dims."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine

def timed(fn, reps=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for name, nx, nu, hidden, H, integ, DT, dt, B in (("c2", 2, 1, [64, 64], 20, "discret", 1.0, torch.float64, 1024),
                                                  ("c5", 2, 1, [64, 64], 50, "discret", 1.0, torch.float64, 1024),
                                                  ("c3", 6, 3, [128] * 3, 30, "rk4", 0.1, torch.float32, 1024)):
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    for kernel in ("mfma", "mfma_tile"):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B, kernel=kernel)
        Z, X0 = (eng.to_device(a) for a in orc.synthetic_inputs(B, H, nx, nu, seed=1))
        full = timed(eng.bind(Z, X0, ("g", "jac_tiles"))[0])
        k_full = eng.last_row_kernel
        only = timed(eng.bind(Z, X0, ("g",))[0])
        print(f"{name} {kernel:9s}: full {full:8.1f} us ({k_full}), defects only {only:8.1f} us ({eng.last_row_kernel})")
