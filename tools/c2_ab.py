#!/usr/bin/env python3
"""The headline (2/1, 2x64 tanh, H=20, Discret, fp64) under several builds of the library, interleaved rounds in one process
per build: HIP-event time of the fused launch at B = 1024 / 256 / 8192 and of the sparse contract, output hashes.
   python tools/c2_ab.py <libA.so> <libB.so> ...      (rounds: ROUNDS=3)"""
import os, subprocess, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for rnd in range(int(os.environ.get("ROUNDS", 3))):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
            print(rnd, os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, REPO)
import numpy as np, torch, hashlib
from pyneuralempc_amd import _lib
_lib.LIB_PATH = sys.argv[2]; _lib._lib = None
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
H, nx, nu = 20, 2, 1
net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
res = {}
for B in (1024, 256, 8192):
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    Z, X0 = eng.to_device(Z), eng.to_device(X0)
    for want in ((("f", "grad", "g", "jac_dense"), ("f", "grad", "g", "jac_sparse")) if B == 1024 else (("f", "grad", "g", "jac_dense"),)):
        step, outs = eng.bind(Z, X0, want)
        reps = 2000 if B <= 1024 else 200
        for _ in range(200): step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): step()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        h = hashlib.sha1(b"".join(outs[k].cpu().numpy().tobytes() for k in want)).hexdigest()[:8]
        res[f"{B}:{want[-1][4:]}"] = (round(best, 3), h)
    del eng
print(json.dumps(res))
