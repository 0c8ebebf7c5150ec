#!/usr/bin/env python3
"""Quick A/B timing of the headline launches (HIP events, primed): C2 fused / unfused at B=1024 and 256, C5 fused (in
place and rotating over buffers larger than the Infinity Cache), Hessian callbacks.  usage: tools/quick_bench.py [tag]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine


def timed(fn, reps=400, prime_ms=40):
    t0 = time.perf_counter(); fn()
    while (time.perf_counter() - t0) * 1e3 < prime_ms:
        for _ in range(16): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(H, B, box, hess=True):
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, H, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=B)
    if box: eng.set_box_rows(-2.0, 2.0)
    Zh, X0h = orc.synthetic_inputs(B, H, 2, 1, seed=1)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    out = {}
    step, _ = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense")); out["fused_us"] = timed(step)
    step, _ = eng.bind(Z, X0, ("g", "jac_tiles")); out["rows_us"] = timed(step)
    jb = B * eng.m * eng.n * 8
    if jb * 3 >= (256 << 20):
        nbuf = max(2, -(-2 * (256 << 20) // jb))
        ring = [eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"), out={"jac_dense": torch.empty(B, eng.m, eng.n, dtype=torch.float64, device="cuda:0")})[0] for _ in range(nbuf)]
        k = [0]
        def rot():
            ring[k[0] % nbuf](); k[0] += 1
        out["fused_rotating_us"] = timed(rot, 200)
    if hess:
        lam = torch.randn(B, eng.m, dtype=torch.float64, device="cuda:0"); sig = torch.ones(B, dtype=torch.float64, device="cuda:0")
        out["hess_us"] = timed(lambda: eng.hess(Z, X0, lam, sig), 200)
        out["gn_us"] = timed(lambda: eng.hess_gn(Z, X0, None, sig), 200)
    return {k: round(v, 2) for k, v in out.items()}


def run_c3(B=1024):
    net = orc.MLP.random(9, [128, 128, 128], 6, seed=0)
    eng = CallbackEngine(net.W, net.b, 30, 6, 3, integrator="rk4", DT=0.1, dtype=torch.float32, device="cuda:0", max_batch=B)
    Zh, X0h = orc.synthetic_inputs(B, 30, 6, 3, seed=1)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    out = {}
    step, _ = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense")); out["eval_us"] = timed(step, 60, 100)
    step, _ = eng.bind(Z, X0, ("g", "jac_tiles")); out["rows_us"] = timed(step, 60, 100)
    lam = torch.randn(B, eng.m, dtype=torch.float32, device="cuda:0"); sig = torch.ones(B, dtype=torch.float32, device="cuda:0")
    out["hess_us"] = timed(lambda: eng.hess(Z, X0, lam, sig), 20, 100)
    return {k: round(v, 1) for k, v in out.items()}


if __name__ == "__main__":
    res = {"c2_b1024": run(20, 1024, False), "c2_b256": run(20, 256, False), "c5": run(50, 1024, True),
           "c2_b4096": run(20, 4096, False, hess=False), "c3": run_c3()}
    print(json.dumps(res))
