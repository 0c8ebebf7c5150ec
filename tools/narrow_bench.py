#!/usr/bin/env python3
"""Narrow networks (width <= 128) on the register-resident kernels vs the layered path: HIP-event time of the rows launch
(g + tiles) and of the exact-Hessian callback, fraction of the matrix peak by (1 + nx) / (2 + nin) network passes.
   python tools/narrow_bench.py"""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine

def timed(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

B, H, nx, nu = 1024, 20, 2, 1
cases = [("3x128 tanh", [128] * 3, "tanh"), ("3x128 mix", [128] * 3, ["relu", "tanh", "sigmoid", "linear"]),
         ("3x64 tanh", [64] * 3, "tanh"), ("4x64 tanh", [64] * 4, "tanh"), ("2x128 tanh", [128] * 2, "tanh"),
         ("2x128 mix", [128] * 2, ["tanh", "softplus", "linear"]), ("2x64 mix", [64] * 2, ["tanh", "elu", "linear"])]
for name, hidden, acts in cases:
    for dt in (torch.float64, torch.float32):
        for kern in ("auto", "layered"):
            net = orc.MLP.random(nx + nu, hidden, nx, seed=0, activations=acts)
            try:
                eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=dt, device="cuda:0", max_batch=B, kernel=kern, activations=net.act)
            except Exception as e:
                print(name, dt, kern, "refused:", str(e)[:80]); continue
            Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            lam, sig = eng.to_device(np.random.default_rng(7).normal(size=(B, eng.m))), eng.to_device(np.ones(B))
            step, _ = eng.bind(Z, X0, ("g", "jac_tiles"))
            t = timed(step, 50)
            ch, _ = eng.bind_hess(Z, X0, lam, sig)
            th = timed(ch, 20)
            dims = [nx + nu] + hidden + [nx]
            F = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:]))
            peak = 78.6 if dt == torch.float64 else 157.3
            fr = B * H * (1 + nx) * F / t / 1e12 / peak
            fh = B * H * (2 + nx + nu) * F / th / 1e12 / peak
            print(f"{name:12s} {'f64' if dt == torch.float64 else 'f32'} {kern:8s} variant={eng.kernel_variant:8s} rows {t*1e6:8.1f} us frac {fr:.3f} [{eng.last_row_kernel}]  hess {th*1e6:8.1f} us frac {fh:.3f} [{eng.last_hess_kernel}]", flush=True)
            del eng
