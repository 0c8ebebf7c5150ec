import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B=1024
for cfgname in ("c2","c3"):
    if cfgname == "c2":
        nx, nu, H, hidden, kind, DT, dt = 2, 1, 20, [64, 64], "discret", 1.0, torch.float64
    else:
        nx, nu, H, hidden, kind, DT, dt = 6, 3, 30, [128, 128, 128], "rk4", 0.1, torch.float32
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=kind, DT=DT, dtype=dt, device="cuda:0", max_batch=B)
    X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    for mu0 in (1.0, 0.1, 0.01, 0.001):
        for mf in (0.2, 0.05):
            eng.solve(X0, lb=lb, ub=-lb, max_iter=3)
            torch.cuda.synchronize(); t=time.perf_counter()
            Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=40, return_iterations=True, mu_init=mu0, mu_factor=mf)
            torch.cuda.synchronize(); dtm=time.perf_counter()-t
            ok=(st==0); p=per[ok].cpu().numpy()
            print(f"{cfgname} mu_init={mu0} mu_factor={mf}: {int(ok.sum())}/{B} in {dtm*1e3:.1f} ms, p50/p95 = {np.percentile(p,50):.0f}/{np.percentile(p,95):.0f}")
