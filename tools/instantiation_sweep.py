#!/usr/bin/env python3
"""Every matrix-core row-kernel instantiation once: dtype x padded width {32, 64, 128} x hidden layers {1, 2, 3} x
transcription x activation x kernel family, defects and Jacobian (and the Lagrangian Hessian values) against the oracle.
One-off confidence run on the GPU box.   python tools/instantiation_sweep.py"""
import os, sys, itertools, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
kinds = {"discret": orc.DISCRET, "rk4": orc.RK4}
H, B = 7, 11
bad = n = 0
for dt, width, depth, integ, act, kern, (nx, nu) in itertools.product(
        (torch.float32, torch.float64), (24, 48, 96), (1, 2, 3), ("discret", "rk4"), ("tanh", "relu", "sigmoid", "softplus", "elu"),
        ("mfma_tile", "mfma"), ((2, 1), (6, 3))):
    DT = 0.1 if integ == "rk4" else 1.0
    net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=5, activations=act)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    prob = orc.Problem(net, H, nx, nu, kinds[integ], DT)
    f, grad, g, J = prob.eval_batch(Zh, X0h)
    try:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B, kernel=kern, activations=act)
    except Exception as e:
        print("create failed", dt, width, depth, integ, act, kern, nx, str(e)[:60]); continue
    res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
    k1 = eng.last_row_kernel
    tol = 2e-4 if dt == torch.float32 else 1e-10
    eg = np.abs(res["g"] - g).max() / max(1, np.abs(g).max()); ej = np.abs(res["jac_dense"] - J).max() / max(1, np.abs(J).max())
    rng = np.random.default_rng(1)
    lam, sig = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(sig))["hvals"].cpu().double().numpy()
    ref = np.stack([prob.hessian_values(Zh[i], X0h[i], lam[i], sig[i]) for i in range(B)])
    eh = np.abs(hv - ref).max() / max(1, np.abs(ref).max())
    n += 1
    if not max(eg, ej) < tol or not eh < (2e-3 if dt == torch.float32 else 1e-9):
        bad += 1
        print(f"BAD {str(dt)[6:]} width={width} depth={depth} {integ} {act} {kern} nx={nx}: g {eg:.1e} jac {ej:.1e} hess {eh:.1e} [{k1}]", flush=True)
    del eng
print("instantiations", n, "bad", bad)
