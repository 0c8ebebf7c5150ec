import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B=1024; nx,nu,H=2,1,20
net = orc.MLP.random(3, [64,64], 2, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
lb = np.concatenate([np.full(H*nx,-3.0), np.full(H*nu,-0.5)])
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5,0.5,size=(B,nx)))
for ce in (2, 4, 8, 1000):
    for compact in (True, False):
        eng.solve(X0, lb=lb, ub=-lb, max_iter=5)
        torch.cuda.synchronize()
        ts=[]
        for _ in range(5):
            t0=time.perf_counter(); Z,st,it = eng.solve(X0, lb=lb, ub=-lb, max_iter=40, check_every=ce, compact=compact); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
        print(f"check_every={ce} compact={compact}: {min(ts):.2f} ms, {it} its, {int((st==0).sum())} converged")
