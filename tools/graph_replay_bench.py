#!/usr/bin/env python3
"""hipGraph replay of one batched evaluation (row kernel + post kernel) vs direct launches: microseconds per evaluation.
The evaluation reads Z / X0 and writes its outputs in fixed device buffers (CallbackEngine.bind), so it can be
captured once and replayed."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine

def timed(fn, reps=300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for name, nx, nu, hidden, H, B in (("c2", 2, 1, [64, 64], 20, 1024), ("c2_b256", 2, 1, [64, 64], 20, 256), ("c5", 2, 1, [64, 64], 50, 1024)):
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B)
    if name == "c5":
        eng.set_box_rows(-2.0, 2.0)
    Z, X0 = (eng.to_device(a) for a in orc.synthetic_inputs(B, H, nx, nu, seed=1))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step, out = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))   # bound to the capture stream
        for _ in range(3): step()
        torch.cuda.synchronize()
        t_direct = timed(step)
        for k in (1, 8):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(k): step()
            t_graph = timed(g.replay) / k
            print(f"{name}: direct {t_direct:6.2f} us/eval, graph of {k}: {t_graph:6.2f} us/eval")
