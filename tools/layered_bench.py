#!/usr/bin/env python3
"""Rows (g + compact tiles) of networks outside the register-resident kernels: the layer-at-a-time GEMM pipeline against the
thread-per-row kernel, B = 1024, fraction of the matrix peak.   python tools/layered_bench.py"""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine

def timed(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

cases = [("wide256_c2", 2, 1, [256, 256], 20, "discret", "tanh"), ("deep4_c2", 2, 1, [64] * 4, 20, "discret", "tanh"),
         ("wide256x3_c2", 2, 1, [256] * 3, 20, "discret", "tanh"), ("wide512x4_c3", 6, 3, [512] * 4, 30, "rk4", "tanh"),
         ("mixed_128x3_c2", 2, 1, [128] * 3, 20, "discret", ["relu", "tanh", "sigmoid", "linear"])]
out = {}
only = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. wide256_c2/float64 (profiling runs)
for name, nx, nu, hidden, H, integ, acts in cases:
    for dt, peak in ((torch.float64, 78.6), (torch.float32, 157.3)):
        if only and only != f"{name}/{str(dt)[6:]}":
            continue
        B = 1024
        DT = 0.1 if integ == "rk4" else 1.0
        net = orc.MLP.random(nx + nu, hidden, nx, seed=0, activations=acts)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
        dims = [nx + nu] + hidden + [nx]
        F = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:]))
        flops = B * H * (4 if integ == "rk4" else 1) * (1 + nx) * F
        row = {}
        for kern in ("layered", "valu"):
            if kern == "valu" and (only or max(hidden) > 256 or integ == "rk4"):
                continue
            eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B, kernel=kern,
                                 activations=net.act)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            step, _ = eng.bind(Z, X0, ("g", "jac_tiles"))
            t = timed(step, 20 if kern == "layered" else 3)
            row[kern] = {"us": t * 1e6, "tflops": flops / t / 1e12, "frac_of_matrix_peak": flops / t / 1e12 / peak, "kernel": eng.last_row_kernel}
            full, _ = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
            row[kern]["dense_eval_us"] = timed(full, 10 if kern == "layered" else 2) * 1e6
            if True:
                # the Lagrangian-Hessian callback (tril values): (2 + nin) GEMM sweeps + the layer-wise contraction (RK4: per
                # stage, inside the stage pipeline, after a rows launch that writes the stage records)
                lam = eng.to_device(np.random.default_rng(2).normal(size=(B, eng.m)))
                sig = eng.to_device(np.ones(B))
                hfn = lambda: eng.hess(Z, X0, lam, sig)
                th = timed(hfn, 10 if kern == "layered" else 2)
                hflops = flops * (2 + nx + nu) / (1 + nx) + (flops if integ == "rk4" else 0)
                row[kern]["hess_us"] = th * 1e6
                row[kern]["hess_kernel"] = eng.last_hess_kernel
                row[kern]["hess_frac_of_matrix_peak"] = hflops / th / 1e12 / peak
            del eng
        out[f"{name}/{str(dt)[6:]}"] = {"gflop": flops / 1e9, **row}
        print(name, str(dt)[6:], json.dumps(row), flush=True)
if not only:
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_layered_bench.json"), "w"), indent=1)
