"""Random sweep of horizons / batch sizes / bounds for the batched solver on the compiled 2/1 shape: the default iteration
(parallel-in-time LQ solve, blocks + evaluation in one launch, acceptance inside the next LQ kernel) against the round-2
machinery (Riccati sweep, separate launches) -- statuses, per-problem iteration counts, solutions.  One-off confidence run
on the GPU box.   python tools/random_solver_sweep.py [trials]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "2024")))
net = orc.MLP.random(3, [64, 64], 2, seed=0)
bad = 0
for trial in range(trials):
    H = int(rng.integers(1, 64)); B = int(rng.integers(1, 300)); integ = ["discret", "unity"][int(rng.integers(0, 2))]
    box = (-2.5, 2.5) if rng.random() < 0.3 else None
    bounded = rng.random() < 0.7
    mi = int(rng.integers(1, 90))
    compact = bool(rng.random() < 0.7)
    if integ == "unity":
        net_u = orc.MLP(net.W, net.b); netx = net_u
    eng = CallbackEngine(net.W, net.b, H, 2, 1, integrator=integ, dtype=torch.float64, device="cuda:0", max_batch=B)
    if box: eng.set_box_rows(*box)
    eng.set_objective(Q=np.eye(2), R=0.1 * np.eye(1))
    lb = np.concatenate([np.full(H * 2, -3.0), np.full(H, -0.5)]) if bounded else None
    ub = None if lb is None else -lb
    X0 = eng.to_device(rng.uniform(-0.5, 0.5, size=(B, 2)))
    new = eng.solve(X0, lb=lb, ub=ub, max_iter=mi, compact=compact, return_iterations=True)
    os.environ["NEMPC_SOLVER_HESS_TRIAL"] = "0"
    old = eng.solve(X0, lb=lb, ub=ub, max_iter=mi, compact=compact, return_iterations=True, lq_kernel="thread")
    del os.environ["NEMPC_SOLVER_HESS_TRIAL"]
    sn, so = new[1].cpu().numpy(), old[1].cpu().numpy()
    conv = (sn == 0) & (so == 0)
    dz = np.abs(new[0].cpu().numpy()[conv] - old[0].cpu().numpy()[conv]).max() if conv.any() else 0.0
    its_eq = np.array_equal(new[3].cpu().numpy()[conv], old[3].cpu().numpy()[conv])
    ok = np.array_equal(sn, so) and dz < 1e-8 and its_eq
    if not ok:
        bad += 1
        print(f"trial {trial}: H={H} B={B} {integ} box={box} bounded={bounded} max_iter={mi} compact={compact}: status equal "
              f"{int((sn == so).sum())}/{B}, max |dZ| {dz:.2e}, iteration counts equal {its_eq}", flush=True)
print("trials", trials, "problems", bad)
