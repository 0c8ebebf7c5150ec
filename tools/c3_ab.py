#!/usr/bin/env python3
"""configs[2] (6/3, 3x128 tanh, H=30, RK4, fp32, B=1024) under two builds of the library: outputs bit for bit, HIP-event time.
   python tools/c3_ab.py <libA.so> <libB.so>"""
import os, subprocess, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] != "--one":
    outs = []
    for lib in sys.argv[1:3]:
        r = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
        print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:])
    sys.exit(0)
sys.path.insert(0, REPO)
import numpy as np, torch, hashlib
from pyneuralempc_amd import _lib
_lib.LIB_PATH = sys.argv[2]; _lib._lib = None
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
B, H, nx, nu = int(os.environ.get("B", 1024)), 30, 6, 3
net = orc.MLP.random(nx + nu, [128, 128, 128], nx, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, dtype=torch.float32, device="cuda:0", max_batch=B)
Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
Z, X0 = eng.to_device(Z), eng.to_device(X0)
res = {}
for want in (("f", "grad", "g", "jac_dense"), ("g", "jac_tiles"), ("f", "grad", "g", "jac_sparse")):
    step, outs = eng.bind(Z, X0, want)
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha1(b"".join(outs[k].cpu().numpy().tobytes() for k in want)).hexdigest()[:12]
    res["+".join(want)] = (round(e0.elapsed_time(e1) * 10, 2), h, eng.last_row_kernel)
print(json.dumps(res))
