"""Random sweep of horizons / batch sizes / transcriptions / box rows for the 2-state 1-control shape against the CPU oracle
(one-off confidence run on the GPU box; the committed tests cover fixed grids).   python tools/random_parity_sweep.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "123")))
net = orc.MLP.random(3, [64, 64], 2, seed=7)
bad = 0
for trial in range(120):
    H = int(rng.integers(1, 90)); B = int(rng.integers(1, 400)); integ = ["discret", "unity"][int(rng.integers(0, 2))]
    box = (-1.0, 1.0) if rng.random() < 0.4 else None
    eng = CallbackEngine(net.W, net.b, H, 2, 1, integrator=integ, dtype=torch.float64, device="cuda:0", max_batch=B)
    if box: eng.set_box_rows(*box)
    eng.set_objective(Q=[[1.0, 0.1], [0.1, 0.5]], R=[[0.2]], cx=0.03, cu=-0.1)
    Zh, X0h = orc.synthetic_inputs(B, H, 2, 1, seed=trial)
    prob = orc.Problem(net, H, 2, 1, orc.UNITY if integ == "unity" else orc.DISCRET, Q=np.array([[1.0, 0.1], [0.1, 0.5]]), R=np.array([[0.2]]),
                       cx=np.full((H, 2), 0.03), cu=np.full((H, 1), -0.1), box=box)
    f, grad, g, J = prob.eval_batch(Zh, X0h)
    res = eng.eval_numpy(Zh, X0h)
    errs = [np.abs(res["f"] - f).max(), np.abs(res["grad"] - grad).max(), np.abs(res["g"] - g).max(), np.abs(res["jac_dense"] - J).max()]
    ok = max(errs) < 1e-11
    if not ok:
        bad += 1
        print("MISMATCH", H, B, integ, box, eng.last_row_kernel, errs)
    del eng
print("trials 120, mismatches", bad)
