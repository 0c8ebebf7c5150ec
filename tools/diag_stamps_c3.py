#!/usr/bin/env python3
"""Per-phase cycle stamps of the configs[2] launch (rows_coop_kernel<float, 128, 3, 1, ...>, 6/3, RK4): diagnostic build of
the fp32 tanh unit with -DNEMPC_STAMPS (never shipped), one evaluation, then per wave of workgroup 0 the cycles between
consecutive stamps of its FIRST pass, stage by stage.   python tools/diag_stamps_c3.py [--build-only] [-DNAME=VALUE ...]"""
import ctypes, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyneuralempc_amd import _build, _lib

def build():
    out = os.path.join(_build.PKG, "build_stamps_c3")
    os.makedirs(out, exist_ok=True)
    _build.build(verbose=False)
    stamped = {"nempc_api.hip", "kernels_mfma_f32.hip"}
    from concurrent.futures import ThreadPoolExecutor
    defs = ["-DNEMPC_STAMPS", "-DNEMPC_STAMPS_NO_FX"] + [f for f in sys.argv if f.startswith("-D")]

    def one(src):
        if src not in stamped:
            return os.path.join(_build.PKG, "build", src.replace(".hip", ".o"))
        o = os.path.join(out, src.replace(".hip", ".o"))
        _build.compile_unit(os.path.join(_build.CSRC, src), o, _build.EXTRA_FLAGS.get(src, []) + defs)
        return o
    with ThreadPoolExecutor(max_workers=2) as ex:
        objs = list(ex.map(one, _build.SOURCES))
    lib = os.path.join(out, "libnempc_stamps_c3.so")
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
    return lib

if __name__ == "__main__":
    if "--build-only" in sys.argv:
        print(build()); sys.exit(0)
    import torch
    from oracle import nempc_oracle as orc
    _lib.LIB_PATH = os.path.join(_build.PKG, "build_stamps_c3", "libnempc_stamps_c3.so")
    _lib._lib = None
    from pyneuralempc_amd import CallbackEngine
    B, H, nx, nu = 1024, 30, 6, 3
    net = orc.MLP.random(nx + nu, [128, 128, 128], nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, dtype=torch.float32, device="cuda:0", max_batch=B)
    eng.lib.nempc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.nempc_debug_stamps(eng._handle, None)
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    Z, X0 = eng.to_device(Z), eng.to_device(X0)
    want = ("g", "jac_tiles") if "tiles" in sys.argv else ("f", "grad", "g", "jac_dense")
    for _ in range(3):
        eng.eval(Z, X0, want)
    print("row kernel:", eng.last_row_kernel)
    buf = np.zeros(1024 + 4096 * 16, dtype=np.int64)
    eng.lib.nempc_debug_stamps(eng._handle, buf.ctypes.data_as(ctypes.c_void_p))
    st = buf[:1024].reshape(16, 64)
    names = {4: "layer0", 5: "hidden fwd", 6: "output+d1", 7: "reverse", 8: "barrier", 10: "reduce", 11: "rk4 chain"}
    for w in range(8):
        s = st[w]
        if s[48] == 0:
            continue
        print(f"wave {w}: pass {int(s[50] - s[48])} cycles; kernel entry -> pass start {int(s[48] - s[0])}; outputs {int(s[50] - s[49])}")
        prev = s[48]
        for stage in range(4):
            parts = []
            for k in (4, 5, 6, 7, 8, 10, 11):
                v = s[12 * stage + k]
                if v:
                    parts.append(f"{names[k]} {int(v - prev)}")
                    prev = v
            print(f"   stage {stage}: " + ", ".join(parts))
    wg = buf[1024:].reshape(4096, 16)
    live = wg[:, 0] != 0
    wg = wg[live]
    life = (wg[:, 14] - wg[:, 0])
    print(f"workgroups {len(wg)}: lifetime cycles min {life.min()} median {int(np.median(life))} max {life.max()}")
