#!/usr/bin/env python3
"""How long the host waits for ONE small evaluation (B = 1, the drop-in path's callback) with three ways of waiting:
stream.synchronize(), spinning on event.query(), spinning on a pinned word the stream writes after the kernel
(hipStreamWriteValue32).   python tools/host_sync_probe.py"""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
H, nx, nu = 10, 2, 1
net = orc.MLP.random(3, [30, 30], 2, seed=0)
eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=1)
Zh, X0h = orc.synthetic_inputs(1, H, nx, nu, seed=1)
Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
stream = torch.cuda.Stream("cuda:0")
with torch.cuda.stream(stream):
    step, out = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWriteValue32.restype = ctypes.c_int
flag = torch.zeros(16, dtype=torch.int32, pin_memory=True)
fnp = flag.numpy()
fptr = ctypes.c_void_p(flag.data_ptr())
sp = ctypes.c_void_p(stream.cuda_stream)
N = 2000
def run(mode):
    ts = []
    seq = 0
    for i in range(N + 50):
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            step()
        if mode == "sync":
            stream.synchronize()
        elif mode == "event":
            ev = torch.cuda.Event(); ev.record(stream)
            while not ev.query(): pass
        else:
            seq += 1
            rc = hip.hipStreamWriteValue32(sp, fptr, seq, 0)
            if rc: raise RuntimeError(f"hipStreamWriteValue32 -> {rc}")
            while fnp[0] != seq: pass
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[50:]) * 1e6
    return np.percentile(ts, [10, 50, 90])
for mode in ("sync", "event", "write32", "sync"):
    try:
        print(mode, "us p10/p50/p90:", run(mode))
    except Exception as e:
        print(mode, "failed:", e)
