#!/usr/bin/env python3
"""One problem's solver iterations (NEMPC_SOLVER_TRACE=<slot> python tools/trace_solver.py [linesearch]): the unbounded
C2-dims test problem family of tests/test_gpu_solver.py."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B = 48; nx, nu, H = 2, 1, 20
net = orc.MLP.random(3, [64, 64], 2, seed=0); net.W[-1] *= 0.2; net.b[-1] *= 0.2
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
X0 = eng.to_device(np.random.default_rng(11).uniform(-1.0, 1.0, size=(B, nx)))
Z, st, it = eng.solve(X0, max_iter=int(os.environ.get("ITERS", "60")), linesearch=sys.argv[1] if len(sys.argv) > 1 else "auto")
print(st.cpu().numpy(), it)
