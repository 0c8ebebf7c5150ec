#!/usr/bin/env python3
"""Rolling-window and extra-input shapes across the matrix-core row / Hessian instantiations: dtype x padded width x hidden
layers x window x activation x kernel family, defects, dense Jacobian and Lagrangian Hessian values against the oracle.
One-off confidence run on the GPU box.   python tools/window_sweep.py"""
import os, sys, itertools, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
H, B = 6, 9
bad = n = 0
rng = np.random.default_rng(3)
for dt, width, depth, w, ne, act, kern, (nx, nu) in itertools.product(
        (torch.float64, torch.float32), (24, 48, 96), (1, 2, 3), (1, 2, 4), (0, 2), ("tanh", "relu"), ("mfma_tile", "mfma"), ((2, 1), (3, 2))):
    if w == 1 and ne == 0:
        continue                                   # (plain shapes: tools/instantiation_sweep.py)
    tw = w * (nx + nu)
    if tw + ne > 32 or (kern == "mfma" and False):
        continue
    net = orc.MLP.random(tw + ne, [width] * depth, nx, seed=7, activations=act)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=7)
    hx, hu = rng.normal(size=(B, w - 1, nx)), rng.uniform(-1, 1, size=(B, w - 1, nu))
    ex = rng.normal(size=(H, ne)) if ne else None
    try:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=dt, device="cuda:0", max_batch=B, kernel=kern, n_extra=ne, rolling_window=w, activations=act)
    except Exception as e:
        print("create failed", dt, width, depth, w, ne, act, kern, nx, str(e)[:70]); continue
    if ne: eng.bind_extra(eng.to_device(np.broadcast_to(ex[None], (B, H, ne)).copy()))
    if w > 1: eng.bind_history(eng.to_device(hx), eng.to_device(hu))
    lam, sig = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
    res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
    k1 = eng.last_row_kernel
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(sig))["hvals"].cpu().double().numpy()
    eg = ej = eh = 0.0
    for i in range(B):
        prob = orc.Problem(net, H, nx, nu, orc.DISCRET, extra=ex, window=w, hist_x=hx[i] if w > 1 else None, hist_u=hu[i] if w > 1 else None)
        g, J = prob.constraints(Zh[i], X0h[i]), prob.jacobian(Zh[i], X0h[i])
        ref = prob.hessian_values(Zh[i], X0h[i], lam[i], sig[i])
        eg = max(eg, np.abs(res["g"][i] - g).max() / max(1, np.abs(g).max()))
        ej = max(ej, np.abs(res["jac_dense"][i] - J).max() / max(1, np.abs(J).max()))
        eh = max(eh, np.abs(hv[i] - ref).max() / max(1, np.abs(ref).max()))
    n += 1
    tol = 2e-4 if dt == torch.float32 else 1e-10
    if not max(eg, ej) < tol or not eh < (5e-3 if dt == torch.float32 else 1e-9):
        bad += 1
        print(f"BAD {str(dt)[6:]} width={width} depth={depth} w={w} ne={ne} {act} {kern} nx={nx}: g {eg:.1e} jac {ej:.1e} hess {eh:.1e} [{k1}]", flush=True)
    del eng
print("shapes", n, "bad", bad)
