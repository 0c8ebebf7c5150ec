#!/usr/bin/env python3
"""Two handles on two HIP streams evaluating independent batches alternately vs one handle on one stream:
microseconds per batched evaluation (C2 dims, B=1024; C5)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
for name, H, box in (("c2", 20, None), ("c5", 50, (-2.0, 2.0))):
    nx, nu, B = 2, 1, 1024
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    engs, steps, streams = [], [], []
    for i in range(2):
        e = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B)
        if box: e.set_box_rows(*box)
        Z, X0 = (e.to_device(a) for a in orc.synthetic_inputs(B, H, nx, nu, seed=1 + i))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            step, _ = e.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
        engs.append((e, Z, X0)); steps.append(step); streams.append(st)
    torch.cuda.synchronize()
    def run(n, two):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(n):
            steps[k % 2 if two else 0]()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6
    run(50, True); run(50, False)
    print(f"{name}: one stream {run(400, False):6.2f} us/eval, two streams alternating {run(400, True):6.2f} us/eval")
