"""Random sweep of shapes for the Lagrangian-Hessian and Gauss-Newton-Hessian callbacks against the CPU oracle (one-off
confidence run on the GPU box).   python tools/random_hessian_sweep.py [trials]"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "77")))
bad = 0
kinds = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}
for trial in range(trials):
    nx = int(rng.integers(1, 5)); nu = int(rng.integers(1, 3))
    depth = int(rng.integers(1, 3)); width = int(rng.choice([8, 16, 32, 64]))
    H = int(rng.integers(1, 14)); B = int(rng.integers(1, 6))
    integ = ["discret", "unity", "rk4"][int(rng.integers(0, 3))]
    kernel = ["auto", "mfma", "mfma_tile", "valu"][int(rng.integers(0, 4))]
    DT = 0.1 if integ == "rk4" else 1.0
    net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=trial)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B,
                         kernel=kernel)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=trial)
    prob = orc.Problem(net, H, nx, nu, kinds[integ], DT)
    lam = rng.standard_normal((B, eng.m)); sig = rng.uniform(0.2, 1.5, size=B)
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(sig))["hvals"].cpu().numpy()
    ref = np.stack([prob.hessian_values(Zh[i], X0h[i], lam[i], sig[i]) for i in range(B)])
    e1 = np.abs(hv - ref).max() / max(1.0, np.abs(ref).max())
    w = rng.uniform(0.0, 2.0, size=(B, H * nx))
    gv = eng.hess_gn(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(w), eng.to_device(sig))["hvals"].cpu().numpy()
    refg = np.stack([prob.gauss_newton_values(Zh[i], X0h[i], w[i], sig[i]) for i in range(B)])
    e2 = np.abs(gv - refg).max() / max(1.0, np.abs(refg).max())
    if not (e1 < 1e-9 and e2 < 1e-9):
        bad += 1
        print("MISMATCH", nx, nu, depth, width, H, B, integ, kernel, "%.1e %.1e" % (e1, e2))
    del eng
print(f"trials {trials}, problems {bad}")
