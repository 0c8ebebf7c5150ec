#!/usr/bin/env python3
"""Host cost of one bound evaluation call (ctypes -> nempc_eval -> one kernel launch): a batch small enough that the GPU
is never the limit; the timed loop of bench.py cannot go below this per step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = orc.MLP.random(3, [64, 64], 2, seed=0)
eng = CallbackEngine(net.W, net.b, 20, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=B)
Z, X0 = orc.synthetic_inputs(B, 20, 2, 1, seed=1)
Z, X0 = eng.to_device(Z), eng.to_device(X0)
step, outs = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
for _ in range(200): step()
torch.cuda.synchronize()
for n in (1000, 5000):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: {n} calls issued in {(t1 - t0) / n * 1e6:.2f} us per call (host), drained after {(t2 - t0) / n * 1e6:.2f} us per call")
