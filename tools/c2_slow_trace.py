#!/usr/bin/env python3
"""Pick a slowly converging problem of the C2 bench batch and print the solver's per-iteration trace for it."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
tgt = int(sys.argv[1]) if len(sys.argv) > 1 else 100
slot = os.environ.get("NEMPC_SOLVER_TRACE")
nx, nu, H, B = 2, 1, 20, 1024
net = orc.MLP.random(3, [64, 64], 2, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=160, return_iterations=True, compact=False)
per = per.cpu().numpy(); st = st.cpu().numpy()
if slot is None:
    cand = np.nonzero((st == 0) & (np.abs(per - tgt) <= 8))[0]
    print("slots converging near", tgt, ":", cand[:8], "per", per[cand[:8]])
    print("histogram of iterations (converged):", np.histogram(per[st == 0], bins=[0, 10, 20, 30, 40, 60, 80, 120, 161])[0])
else:
    print("slot", slot, "status", st[int(slot)], "iterations", per[int(slot)])
