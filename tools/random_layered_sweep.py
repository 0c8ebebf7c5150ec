#!/usr/bin/env python3
"""Random shapes through the layer-at-a-time GEMM path: rows (g, dense Jacobian) and Lagrangian blocks against the oracle, fp64
and fp32, every integrator, activation mixes incl. swish / gelu, extra inputs, box rows.  NEMPC_SWEEP_SEED picks the seed,
N the number of cases.   python tools/random_layered_sweep.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "0")))
N = int(os.environ.get("N", "40"))
ACTS = ["tanh", "relu", "sigmoid", "softplus", "elu", "elu:0.6", "leaky_relu:0.15", "selu", "swish", "gelu", "softsign", "mish",
        "exponential", "relu6", "linear"]
bad = 0
for case in range(N):
    nx, nu = int(rng.integers(1, 7)), int(rng.integers(1, 5))
    nl_h = int(rng.integers(1, 6))
    hidden = [int(rng.integers(5, 300)) for _ in range(nl_h)]
    integ = ["discret", "unity", "rk4"][int(rng.integers(0, 3))]
    H, B = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    acts = [ACTS[int(rng.integers(0, len(ACTS)))] for _ in range(nl_h)]
    out_act = ["linear", "linear", "tanh", "softplus"][int(rng.integers(0, 4))]
    zb = any(a in orc.ZBASED for a in acts)
    if nl_h == 1 and out_act != "linear" and zb:
        out_act = "linear"           # (the one shape nempc_create refuses for swish / gelu)
    acts = acts + [out_act]
    dt = [torch.float64, torch.float32][int(rng.integers(0, 2))]
    DT = 0.1 if integ == "rk4" else 1.0
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[integ]
    box = (-1.0, 1.0) if integ != "rk4" and rng.integers(0, 2) else None
    net = orc.MLP.random(nx + nu, hidden, nx, seed=int(rng.integers(0, 1000)), activations=acts)
    prob = orc.Problem(net, H, nx, nu, kind, DT, box=box)
    tag = f"case {case}: {nx}/{nu} {hidden} {acts} {integ} H={H} B={B} {str(dt)[6:]} box={box}"
    try:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B, kernel="layered",
                             activations=net.act)
        if box:
            eng.set_box_rows(*box)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=case)
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        k = min(B, 3)
        f, grad, g, J = prob.eval_batch(Zh[:k], X0h[:k])
        tol = 1e-10 if dt == torch.float64 else 2e-3
        scale = max(1.0, np.abs(J).max())
        e1 = max(np.abs(res["g"][:k] - g).max(), np.abs(res["jac_dense"][:k] - J).max()) / scale
        lam = rng.normal(size=(B, eng.m))
        hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.ones(B)))["hvals"].to("cpu", torch.float64).numpy()
        e2 = 0.0
        for i in range(min(B, 2)):
            ref = prob.hessian_values(Zh[i], X0h[i], lam[i], 1.0)
            e2 = max(e2, np.abs(hv[i] - ref).max() / max(1.0, np.abs(ref).max()))
        ok = e1 < tol and e2 < (1e-9 if dt == torch.float64 else 1e-2) and np.isfinite(hv).all()
        print(("ok  " if ok else "BAD ") + tag + f"  rows {e1:.1e} hess {e2:.1e} [{eng.last_hess_kernel}]", flush=True)
        bad += 0 if ok else 1
        del eng
    except Exception as ex:      # noqa
        print("EXC " + tag + f"  {type(ex).__name__}: {ex}", flush=True)
        bad += 1
print("failures:", bad)
