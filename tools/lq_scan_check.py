#!/usr/bin/env python3
"""The parallel-in-time LQ solve against the sweep, inside the batched solver: converged counts, per-problem iteration
counts, solutions and wall time at C2 / C5 dims.   python tools/lq_scan_check.py [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, H, box in (("c2", 20, None), ("c5", 50, (-2.0, 2.0))):
    nx, nu = 2, 1
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    if box: eng.set_box_rows(*box)
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
    res = {}
    for mi in (1, 2, 40, 160):
        for k in ("thread", "scan"):
            eng.solve(X0, lb=lb, ub=-lb, max_iter=5, lq_kernel=k)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                Z, st, it, its = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi, lq_kernel=k, return_iterations=True)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            res[k] = (Z.cpu().numpy(), st.cpu().numpy(), its.cpu().numpy())
            print(f"{name} B={B} max_iter={mi} {k:6s}: {best:.2f} ms, {it} iterations, {int((st == 0).sum())} converged", flush=True)
        (Za, sa, ia), (Zb, sb, ib) = res["thread"], res["scan"]
        both = (sa == 0) & (sb == 0)
        print(f"   status equal {int((sa == sb).sum())}/{B}; both converged {int(both.sum())}: max |dZ| {np.abs(Za[both] - Zb[both]).max() if both.any() else 0:.2e}, "
              f"iteration counts equal {int((ia[both] == ib[both]).sum())}; all problems max |dZ| {np.abs(Za - Zb).max():.2e}", flush=True)
