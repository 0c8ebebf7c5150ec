#!/usr/bin/env python3
"""configs[2] dims, B = 1024: converged problems after 40 / 60 / 80 iterations under settings of the Levenberg term's
raise / relax factors (one process per setting).   python tools/solver_reg_sweep.py "RAISE RELAX" ..."""
import os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for setting in sys.argv[1:]:
        env = dict(os.environ)
        for kv in setting.split():
            k, v = kv.split("=", 1); env[k] = v
        r = subprocess.run([sys.executable, __file__, "--one"], capture_output=True, text=True, env=env)
        print(repr(setting), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
    sys.exit(0)
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
nx, nu, H, hidden = 6, 3, 30, [128, 128, 128]
B = 1024
net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, dtype=torch.float32, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
out = []
for mi in (40, 60, 80):
    eng.solve(X0, lb=lb, ub=-lb, max_iter=3)
    torch.cuda.synchronize(); t = time.perf_counter()
    Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi, return_iterations=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = (st == 0)
    p = per[ok].cpu().numpy()
    out.append(f"{mi}: {int(ok.sum())}/{B} in {dt*1e3:.0f} ms (p50 {np.percentile(p, 50):.0f})")
print("   ".join(out))
