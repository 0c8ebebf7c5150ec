#!/usr/bin/env python3
"""Measurements for the SURVEY 8(f) rows at C2 dims (B=1024, H=20, 2/1 MLP 2x64, fp64): one JSON line per row with
problem-evals/s, microseconds per batched evaluation (HIP events), max abs error against the CPU oracle on a sample, and
the single-core NumPy oracle rate of the same workload (the reference-shaped CPU path) beside it.

  f2 sparse    all four callbacks with the Jacobian in the exact band pattern (jac_sparse) instead of dense (m,n)
  f2 tiles     the compact per-step tile contract
  f3 p_tvp     p_dim=1, tvp_dim=2 extra network inputs
  f4 rolling   rolling window w=2 / w=4 (wave-per-tile matrix-core kernel), incl. the Lagrangian Hessian
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc                      # noqa: E402  (checker / CPU baseline only)
from pyneuralempc_amd import CallbackEngine                  # noqa: E402

B, H, nx, nu, hidden = 1024, 20, 2, 1, [64, 64]
dev = "cuda:0"


def timed(fn, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def cpu_rate(fn, seconds=4.0):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn(n)
        n += 1
    return n / (time.perf_counter() - t0)


def report(row, eng, Z, X0, want, make_prob, Zh, X0h, hess=False):
    step, out = eng.bind(Z, X0, want)
    t = timed(step)
    errs = {}
    res = {k: v[:8].to("cpu", torch.float64).numpy() for k, v in out.items()}
    rows, cols = eng.jac_structure()
    for i in range(8):
        prob = make_prob(i)
        J = prob.jacobian(Zh[i], X0h[i])
        ref = {"f": prob.objective(Zh[i]), "grad": prob.gradient(Zh[i]), "g": prob.constraints(Zh[i], X0h[i]),
               "jac_dense": J, "jac_sparse": J[rows, cols], "jac_tiles": prob.tiles(Zh[i], X0h[i])[2]}
        for k in res:
            errs[k] = max(errs.get(k, 0.0), float(np.abs(res[k][i] - ref[k]).max()))
    prob0 = make_prob(0)

    def cpu_eval(n):
        i = n % 8
        prob0.objective(Zh[i]); prob0.gradient(Zh[i]); prob0.constraints(Zh[i], X0h[i]); prob0.jacobian(Zh[i], X0h[i])
    line = {"row": row, "workload": f"B={B}, H={H}, {nx}/{nu} MLP{hidden}, fp64, outputs {list(want)}",
            "problem_evals_per_s": B / t, "batch_evals_per_s": 1.0 / t, "eval_us": t * 1e6, "row_kernel": eng.last_row_kernel,
            "max_abs_err_vs_cpu_oracle": errs,
            "cpu_baseline": {"value": cpu_rate(cpu_eval), "unit": "problem-evals/s", "cores": 1, "kind": "port",
                             "sample": "NumPy oracle, per-problem dense evaluation, ~4 s"}}
    if hess:
        lam = torch.randn(B, eng.m, dtype=torch.float64, device=dev)
        sig = torch.ones(B, dtype=torch.float64, device=dev)
        th = timed(lambda: eng.hess(Z, X0, lam, sig), 50)
        hv = eng.hess(Z, X0, lam, sig, want=("hdense",))["hdense"][:2].cpu().numpy()
        herr = max(float(np.abs(hv[i] - make_prob(i).lagrangian_hessian(Zh[i], X0h[i], lam[i].cpu().numpy(), 1.0)).max())
                   for i in range(2))
        line["hessian_callback"] = {"hess_us": th * 1e6, "nnz_hess": eng.nnz_hess, "max_abs_err_vs_cpu_oracle": herr}
    print(json.dumps(line), flush=True)


def main():
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, device=dev, max_batch=B)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    plain = lambda i: orc.Problem(net, H, nx, nu, orc.DISCRET)
    report("f2_sparse_contract", eng, Z, X0, ("f", "grad", "g", "jac_sparse"), plain, Zh, X0h)
    report("f2_tile_contract", eng, Z, X0, ("f", "grad", "g", "jac_tiles"), plain, Zh, X0h)
    report("dense_contract(reference)", eng, Z, X0, ("f", "grad", "g", "jac_dense"), plain, Zh, X0h)

    p_dim, tvp_dim = 1, 2
    net3 = orc.MLP.random(nx + nu + p_dim + tvp_dim, hidden, nx, seed=0)
    rng = np.random.default_rng(4)
    extra = np.concatenate([rng.normal(size=(B, H, tvp_dim)), np.tile(rng.normal(size=(B, 1, p_dim)), (1, H, 1))], axis=2)
    eng3 = CallbackEngine(net3.W, net3.b, H, nx, nu, device=dev, max_batch=B, n_extra=p_dim + tvp_dim)
    eng3.bind_extra(eng3.to_device(extra))
    report("f3_p_tvp", eng3, Z, X0, ("f", "grad", "g", "jac_dense"),
           lambda i: orc.Problem(net3, H, nx, nu, orc.DISCRET, extra=extra[i]), Zh, X0h, hess=True)

    for w in (2, 4):
        netw = orc.MLP.random(w * (nx + nu), hidden, nx, seed=0)
        hx, hu = rng.normal(size=(B, w - 1, nx)), rng.uniform(-1, 1, size=(B, w - 1, nu))
        engw = CallbackEngine(netw.W, netw.b, H, nx, nu, device=dev, max_batch=B, rolling_window=w)
        engw.bind_history(engw.to_device(hx), engw.to_device(hu))
        report(f"f4_rolling_window_{w}", engw, Z, X0, ("f", "grad", "g", "jac_dense"),
               lambda i, netw=netw, w=w, hx=hx, hu=hu: orc.Problem(netw, H, nx, nu, orc.DISCRET, window=w, hist_x=hx[i],
                                                                     hist_u=hu[i]), Zh, X0h, hess=True)


if __name__ == "__main__":
    main()
