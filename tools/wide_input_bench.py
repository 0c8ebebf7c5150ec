#!/usr/bin/env python3
"""Rows with 17..32 network inputs (rolling windows of wider systems): wave-per-tile matrix-core kernel vs the generic
kernel, microseconds per batched evaluation (B=1024, H=20, 3/2 MLP 2x64, window 4 -> 20 inputs, fp64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B, H, nx, nu, w = 1024, 20, 3, 2, 4
net = orc.MLP.random(w * (nx + nu), [64, 64], nx, seed=0)
rng = np.random.default_rng(1)
hx, hu = rng.normal(size=(B, w - 1, nx)), rng.uniform(-1, 1, size=(B, w - 1, nu))
Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
def timed(fn, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for kernel in ("mfma", "valu"):
    eng = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B, kernel=kernel, rolling_window=w)
    eng.bind_history(eng.to_device(hx), eng.to_device(hu))
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    t = timed(eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))[0])
    lam = torch.randn(B, eng.m, dtype=torch.float64, device="cuda:0"); sig = torch.ones(B, dtype=torch.float64, device="cuda:0")
    th = timed(lambda: eng.hess(Z, X0, lam, sig), 10)
    print(f"{kernel:5s}: eval {t:8.1f} us ({eng.last_row_kernel}), hessian callback {th:9.1f} us")
