#!/usr/bin/env python3
"""Converged fraction vs the barrier schedule (nempc_solve mu_init, or with SWEEP=factor: mu_factor at mu_init values 0.1 and
0.03), C2 and configs[2] dims.   python tools/solver_mu_init_sweep.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
for cfgname in ("c2", "c3"):
    if cfgname == "c2":
        nx, nu, H, hidden, kind, DT, dt = 2, 1, 20, [64, 64], "discret", 1.0, torch.float64
    else:
        nx, nu, H, hidden, kind, DT, dt = 6, 3, 30, [128, 128, 128], "rk4", 0.1, torch.float32
    B = 1024
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=kind, DT=DT, dtype=dt, device="cuda:0", max_batch=B)
    X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    sweep_factor = os.environ.get("SWEEP") == "factor"
    grid = [(m, f) for m in (1e-1, 3e-2) for f in (0.2, 0.1, 0.05, 0.02)] if sweep_factor else [(m, 0.2) for m in (1e-1, 3e-2, 1e-2, 1e-3)]
    for mu0, fac in grid:
        row = []
        for mi in (40, 60, 160):
            eng.solve(X0, lb=lb, ub=-lb, max_iter=3, mu_init=mu0, mu_factor=fac)
            torch.cuda.synchronize(); t = time.perf_counter()
            Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi, mu_init=mu0, mu_factor=fac, return_iterations=True)
            torch.cuda.synchronize(); dtm = time.perf_counter() - t
            ok = st == 0
            p = per[ok].cpu().numpy()
            row.append(f"{mi}: {int(ok.sum())}/{B} in {dtm*1e3:.1f} ms (p50 {np.percentile(p, 50) if len(p) else 0:.0f})")
        print(cfgname, "mu_init", mu0, "mu_factor", fac, " | ".join(row), flush=True)
