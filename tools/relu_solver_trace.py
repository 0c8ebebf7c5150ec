#!/usr/bin/env python3
"""Per-iteration solver trace of one problem of the relu test batch (tests/test_gpu_solver.py::test_batched_solve_with_other_activations).
   NEMPC_SOLVER_TRACE=<slot> python tools/relu_solver_trace.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
nx, nu, H, B = 2, 1, 12, 32
net = orc.MLP.random(nx + nu, [64, 64], nx, seed=3, activations="relu")
net.W[-1] *= 0.2; net.b[-1] *= 0.2
lb = np.concatenate([np.full(H * nx, -10.0), np.full(H * nu, -0.5)])
X0 = np.random.default_rng(5).uniform(-0.6, 0.6, size=(B, nx))
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, activations="relu")
eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
Z, status, iters = eng.solve(eng.to_device(X0), lb=lb, ub=-lb, max_iter=int(os.environ.get("ITERS", "120")), compact=False)
print("converged", int((status == 0).sum().item()), "of", B, "in", iters)
