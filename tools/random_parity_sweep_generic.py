"""Random sweep of problem shapes (states, controls, widths, depth, horizon, batch, transcription, dtype, kernel family)
against the CPU oracle: a one-off confidence run on the GPU box for the generic kernels (the committed tests cover goldens
and seeded cases).   python tools/random_parity_sweep_generic.py [trials]"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "2024")))
bad = 0
kinds = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}
for trial in range(trials):
    nx = int(rng.integers(1, 7)); nu = int(rng.integers(1, 4))
    depth = int(rng.integers(1, 4)); width = int(rng.choice([8, 16, 24, 32, 48, 64, 96, 128]))
    H = int(rng.integers(1, 40)); B = int(rng.integers(1, 200))
    integ = ["discret", "unity", "rk4"][int(rng.integers(0, 3))]
    kernel = ["auto", "mfma", "mfma_tile", "valu"][int(rng.integers(0, 4))]
    f32 = rng.random() < 0.3
    DT = 0.1 if integ == "rk4" else 1.0
    net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=trial)
    try:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float32 if f32 else torch.float64,
                             device="cuda:0", max_batch=B, kernel=kernel)
    except Exception as e:                      # noqa: BLE001
        print("create failed", nx, nu, depth, width, H, B, integ, kernel, type(e).__name__, str(e)[:80]); bad += 1; continue
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=trial)
    prob = orc.Problem(net, H, nx, nu, kinds[integ], DT)
    f, grad, g, J = prob.eval_batch(Zh, X0h)
    res = eng.eval_numpy(Zh, X0h)
    tol = 2e-4 if f32 else 1e-10
    def rel(a, b): return np.abs(a - b).max() / max(1.0, np.abs(b).max())
    errs = [rel(res["f"], f), rel(res["grad"], grad), rel(res["g"], g), rel(res["jac_dense"], J)]
    if not max(errs) < tol:
        bad += 1
        print("MISMATCH", nx, nu, depth, width, H, B, integ, kernel, "f32" if f32 else "f64", eng.last_row_kernel, ["%.1e" % e for e in errs])
    del eng
print(f"trials {trials}, problems {bad}")
