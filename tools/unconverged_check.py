#!/usr/bin/env python3
"""Which of the problems the batched solver leaves unconverged have a trajectory inside the bounds at all?
For EVERY unconverged problem of a bench batch (configs[4]: B=1024, H=50, states in [-2, 2] from the box rows and |x| <= 3,
|u| <= 0.5; or C2 dims) a bounded nonlinear least-squares fit of the defects on the CPU oracle (scipy least_squares, trf,
analytic Jacobian), started from the solver's last iterate and from the cold start: a residual that stays far from zero
means no feasible trajectory exists within the bounds.   python tools/unconverged_check.py [c5|c2] [max_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ProcessPoolExecutor
from scipy.optimize import least_squares
from oracle import nempc_oracle as orc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 400
H = 50 if cfg == "c5" else 20
B, nx, nu = 1024, 2, 1
net = orc.MLP.random(3, [64, 64], 2, seed=0)
prob = orc.Problem(net, H, nx, nu, orc.DISCRET)
xlim = 2.0 if cfg == "c5" else 3.0
lb = np.concatenate([np.full(H * nx, -xlim), np.full(H * nu, -0.5)])


def fit(args):
    z0, x0 = args
    best = None
    for start in (np.clip(z0, lb + 1e-9, -lb - 1e-9), np.clip(orc.cold_start(x0, H, nu), lb + 1e-9, -lb - 1e-9)):
        r = least_squares(lambda z: prob.constraints(z, x0), start, jac=lambda z: prob.jacobian(z, x0), bounds=(lb, -lb),
                          method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=200)
        res = float(np.abs(r.fun).max())
        best = res if best is None else min(best, res)
    return best


if __name__ == "__main__":
    dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"unconverged_{cfg}.npz")
    if "--fit" not in sys.argv:
        # phase 1 (GPU): solve, keep the iterates; phase 2 runs in a fresh process that never touches the GPU (worker
        # processes must not be forked from one that has)
        from pyneuralempc_amd import CallbackEngine
        eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
        if cfg == "c5":
            eng.set_box_rows(-2.0, 2.0)
        X0h = np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx))
        lbs = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
        Z, st, it = eng.solve(eng.to_device(X0h), lb=lbs, ub=-lbs, max_iter=max_iter)
        np.savez(dump, Z=Z.cpu().numpy(), st=st.cpu().numpy(), X0=X0h, it=it)
        import subprocess
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # one BLAS thread per worker
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), cfg, str(max_iter), "--fit"], env=env))
    d = np.load(dump)
    Zh, st, X0h = d["Z"], d["st"], d["X0"]
    bad = np.nonzero(st != 0)[0]
    good = np.nonzero(st == 0)[0][:8]
    print(f"{cfg}: {len(bad)} of {B} unconverged after {int(d['it'])} iterations", flush=True)
    t0 = time.time()
    res_bad = []
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for k, r in enumerate(ex.map(fit, [(Zh[i], X0h[i]) for i in bad])):
            res_bad.append(r)
            if k % 8 == 7:
                print(f"   fitted {k + 1} / {len(bad)}  [{time.time() - t0:.0f} s]", flush=True)
        res_good = list(ex.map(fit, [(Zh[i], X0h[i]) for i in good]))
    res_bad = np.array(res_bad)
    print(f"bounded least-squares residual max|defect| of the UNCONVERGED problems: min {res_bad.min():.2e} median {np.median(res_bad):.2e} "
          f"max {res_bad.max():.2e}; feasible (< 1e-8): {(res_bad < 1e-8).sum()} of {len(bad)}   [{time.time() - t0:.0f} s]")
    print("   feasible ones:", bad[res_bad < 1e-8].tolist())
    print("   residuals:", np.array2string(np.sort(res_bad), precision=2, max_line_width=200))
    print(f"control: 8 converged problems: max residual {max(res_good):.2e}")
