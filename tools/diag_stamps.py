#!/usr/bin/env python3
"""Diagnostic build with in-kernel s_memtime stamps (never shipped): builds pyneuralempc_amd/build_stamps/
libnempc_stamps.so with -DNEMPC_STAMPS, runs one C2-shaped evaluation and prints, per wave of
workgroup 0, the cycles between consecutive stamps of its LAST pass.  Read the shares, not the total."""
import ctypes, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyneuralempc_amd import _build, _lib

def build():
    out = os.path.join(_build.PKG, "build_stamps")
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in _build.SOURCES:
        o = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.run([_build._hipcc()] + _build.FLAGS + ["-DNEMPC_STAMPS"] + [f for f in sys.argv if f.startswith("-D")] + ["-c", os.path.join(_build.CSRC, src), "-o", o], check=True)
        objs.append(o)
    lib = os.path.join(out, "libnempc_stamps.so")
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib

if __name__ == "__main__":
    if "--build-only" in sys.argv:
        print(build()); sys.exit(0)
    import torch
    from oracle import nempc_oracle as orc
    lib = os.path.join(_build.PKG, "build_stamps", "libnempc_stamps.so")
    _lib.LIB_PATH = lib
    _lib._lib = None
    from pyneuralempc_amd import CallbackEngine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    kernel = sys.argv[2] if len(sys.argv) > 2 else "mfma"
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, 20, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel)
    eng.lib.nempc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.nempc_debug_stamps(eng._handle, None)
    Z, X0 = orc.synthetic_inputs(B, 20, 2, 1, seed=1)
    Z, X0 = eng.to_device(Z), eng.to_device(X0)
    want = ("g", "jac_dense") if (len(sys.argv) > 3 and sys.argv[3] == "dense") else ("g", "jac_tiles")
    for _ in range(3):
        eng.eval(Z, X0, want)
    buf = np.zeros(1024, dtype=np.int64)
    eng.lib.nempc_debug_stamps(eng._handle, buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.reshape(16, 64)
    for w in range(8):
        s = st[w][:13]
        if s[0] == 0: continue
        print(f"wave {w}: " + " ".join(f"{i}:{int(s[i + 1] - s[i]):>6d}" for i in range(12) if s[i] and s[i + 1]) + f"   total {int(s[12]-s[0])}")
