import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B=64; nx,nu,H=2,1,20
net = orc.MLP.random(3, [64, 64], 2, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
Z, st, it = eng.solve(X0, lb=lb, ub=-lb, max_iter=40)
print(st[:8])
