#!/usr/bin/env python3
"""Random shapes through the batched solver (the generic Riccati kernels, both LQ variants, fp64): every problem the solver
reports converged must be feasible and stationary on the oracle's callbacks.   python tools/random_solver_shapes.py [trials]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(os.environ.get("NEMPC_SWEEP_SEED", "11")))
kinds = {"discret": orc.DISCRET, "rk4": orc.RK4}
bad = 0; conv_frac = []
for trial in range(trials):
    nx = int(rng.integers(1, 5)); nu = int(rng.integers(1, 3)); H = int(rng.integers(2, 13)); B = int(rng.integers(1, 12))
    width = int(rng.choice([16, 48, 80])); depth = int(rng.integers(1, 3))
    integ = ["discret", "rk4"][int(rng.integers(0, 2))]; DT = 0.1 if integ == "rk4" else 1.0
    act = ["tanh", "softplus", "sigmoid"][int(rng.integers(0, 3))]
    lqk = ["auto", "thread", "wave"][int(rng.integers(0, 3))]
    kern = ["auto", "valu"][int(rng.integers(0, 2))]
    net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=trial, activations=act)
    net.W[-1] *= 0.2; net.b[-1] *= 0.2
    Q, R = np.eye(nx), 0.1 * np.eye(nu)
    prob = orc.Problem(net, H, nx, nu, kinds[integ], DT, Q=Q, R=R)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kern, activations=act)
    eng.set_objective(Q=Q, R=R)
    n = H * (nx + nu)
    bounded = rng.random() < 0.6
    lb = np.concatenate([np.full(H * nx, -4.0), np.full(H * nu, -0.3)]) if bounded else np.full(n, -np.inf)
    ub = -lb
    X0 = rng.uniform(-0.6, 0.6, size=(B, nx))
    Z, st, it = eng.solve(eng.to_device(X0), lb=lb if bounded else None, ub=ub if bounded else None, max_iter=300, lq_kernel=lqk)
    Z, st = Z.cpu().numpy(), st.cpu().numpy()
    conv_frac.append((st == 0).mean())
    for i in np.nonzero(st == 0)[0]:
        g = prob.constraints(Z[i], X0[i])
        J, gr = prob.jacobian(Z[i], X0[i]), prob.gradient(Z[i])
        free = (Z[i] > lb + 1e-3) & (Z[i] < ub - 1e-3)
        lam = np.linalg.lstsq(J[:, free].T, -gr[free], rcond=None)[0]
        r = np.abs(gr[free] + J[:, free].T @ lam).max() / max(1.0, np.abs(gr).max())
        inb = (Z[i] >= lb - 1e-9).all() and (Z[i] <= ub + 1e-9).all()
        if not (np.abs(g).max() < 1e-6 and r < 1e-4 and inb):
            bad += 1
            print(f"trial {trial} problem {i}: nx={nx} nu={nu} H={H} {integ} {act} w={width}x{depth} lq={lqk} kern={kern}: |g| {np.abs(g).max():.1e} stationarity {r:.1e} in bounds {inb}", flush=True)
    del eng
print("trials", trials, "bad converged problems", bad, "mean converged fraction %.3f, min %.2f" % (np.mean(conv_frac), np.min(conv_frac)))
