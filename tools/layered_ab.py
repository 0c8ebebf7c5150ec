#!/usr/bin/env python3
"""The layered path under run-time switches, one process per setting, interleaved rounds: whole evaluation (f, grad, g, dense
Jacobian) and exact-Hessian callback of 2 x 256 and 3 x 256 tanh networks (2/1, H = 20, B = 1024, fp64), HIP events after 40 ms
of priming launches, best of three (bench.py's layered leg, as an A/B).
   python tools/layered_ab.py "" "NEMPC_LAYERED_DFA=2" ...        (ROUNDS=2)"""
import os, subprocess, sys, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for rnd in range(int(os.environ.get("ROUNDS", 2))):
        for setting in sys.argv[1:]:
            env = dict(os.environ)
            for kv in setting.split():
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([sys.executable, __file__, "--one"], capture_output=True, text=True, env=env)
            print(rnd, repr(setting), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B, H, nx, nu = 1024, 20, 2, 1
res = {}
def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 1)
for name, hidden in (("2x256", [256, 256]), ("3x256", [256] * 3)):
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    rng = np.random.default_rng(7)
    lam, sig = eng.to_device(rng.normal(size=(B, eng.m))), eng.to_device(rng.uniform(0.5, 1.5, size=B))
    step, _ = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
    rows, _ = eng.bind(Z, X0, ("g", "jac_tiles"))
    ch, _ = eng.bind_hess(Z, X0, lam, sig)
    res[name] = {"eval": timed(step, 50), "rows": timed(rows, 50), "hess": timed(ch, 30)}
print(json.dumps(res))
