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
    # only the translation units that see the stamps are rebuilt (the C ABI's debug entry point and the fp64 tanh
    # kernels); the rest come from the main build
    _build.build(verbose=False)
    stamped = {"nempc_api.hip", "kernels_mfma_f64.hip"}
    from concurrent.futures import ThreadPoolExecutor

    def one(src):
        if src not in stamped:
            return os.path.join(_build.PKG, "build", src.replace(".hip", ".o"))
        o = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.run([_build._hipcc()] + _build.FLAGS + _build.EXTRA_FLAGS.get(src, []) + ["-DNEMPC_STAMPS"] +
                       [f for f in sys.argv if f.startswith("-D")] + ["-c", os.path.join(_build.CSRC, src), "-o", o], check=True)
        return o
    with ThreadPoolExecutor(max_workers=2) as ex:
        objs = list(ex.map(one, _build.SOURCES))
    lib = os.path.join(out, "libnempc_stamps.so")
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
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
    Hd = int(os.environ.get("NEMPC_DIAG_H", "20"))
    eng = CallbackEngine(net.W, net.b, Hd, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel)
    if os.environ.get("NEMPC_DIAG_BOX"):
        eng.set_box_rows(-2.0, 2.0)
    eng.lib.nempc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.nempc_debug_stamps(eng._handle, None)
    Z, X0 = orc.synthetic_inputs(B, Hd, 2, 1, seed=1)
    Z, X0 = eng.to_device(Z), eng.to_device(X0)
    mode = sys.argv[3] if len(sys.argv) > 3 else "tiles"
    want = {"dense": ("g", "jac_dense"), "fused": ("f", "grad", "g", "jac_dense")}.get(mode, ("g", "jac_tiles"))
    if mode == "fused":
        eng.set_objective(Q=np.eye(2), R=np.eye(1))
    for _ in range(3):
        eng.eval(Z, X0, want)
    buf = np.zeros(1024 + 4096 * 16, dtype=np.int64)
    eng.lib.nempc_debug_stamps(eng._handle, buf.ctypes.data_as(ctypes.c_void_p))
    st = buf[:1024].reshape(16, 64)
    # ---- timeline of every workgroup (100 MHz ticks -> us), last evaluation
    wg = buf[1024:].reshape(4096, 16)
    live = wg[:, 0] != 0
    wg = wg[live]
    # shader-clock stamps; the one real-time stamp (word 13, taken right after the exit stamp, word 14) puts the
    # workgroups on a common axis: t(event) = real_exit - (clk_exit - clk_event) / f
    span_clk = (wg[:, 14] - wg[:, 0]).astype(np.float64)
    f_mhz = float(os.environ.get("NEMPC_CLK_MHZ", "0")) or None
    if f_mhz is None:
        # clock from the spread of exits: among workgroups of one XCD the shader counter is common
        f_mhz = 2250.0
    real_exit_us = wg[:, 13] / 100.0
    T = np.zeros(wg.shape, dtype=np.float64)            # every stamp of every workgroup on the common axis, us
    for i in range(15):
        T[:, i] = real_exit_us - (wg[:, 14] - wg[:, i]) / f_mhz
    T -= T[:, 0].min()
    T[wg == 0] = np.nan
    hw = wg[:, 15]
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cukey = xcc * 1000 + se * 100 + sh * 10 + cu
    print(f"workgroups {len(wg)}, distinct CUs {len(set(cukey.tolist()))}, XCCs {sorted(set(xcc.tolist()))}")
    print(f"kernel span (first entry -> last exit): {np.nanmax(T[:, 14]):.2f} us   [shader clock assumed {f_mhz:.0f} MHz; "
          f"median workgroup lifetime {np.median(span_clk):.0f} cycles]")
    names = ["entry", "stage_ld issued", "blob->LDS", "weights issued", "p1 staged", "p1 done", "p2 staged", "p2 done",
             "p3 staged", "p3 done"]
    if eng.last_row_kernel == "rows_coopfx_kernel":      # fixed-shape kernel: stamps of the LAST pass of each workgroup
        names = ["entry", "loads issued", "tables in LDS", "weights arrived", "first barrier", "forward done",
                 "reverse done", "partials barrier", "tiles + g out", "-", "-", "non-zeros issued",
                 "pass end"]
        if any(f == "-DNEMPC_STAMPS_PRO" for f in sys.argv) or os.environ.get("NEMPC_STAMPS_PRO"):
            names = ["entry", "inputs issued", "tables issued", "objective loads issued", "weights issued",
                     "tables in LDS", "objective data in LDS", "inputs in LDS", "first barrier", "objective done",
                     "layer 0 done", "hidden forward done", "reverse done"]
    for lo, hi, label in ((0, 256, "wg 0..255 (one tile more)"), (256, 512, "wg 256..511")):
        sel = wg[lo:hi]
        if not len(sel): continue
        print(label)
        for i, nm in enumerate(names):
            col = sel[:, i]
            ok = col != 0
            if ok.any():
                v = T[lo:hi, i][ok]
                print(f"   {nm:>16s}: min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us   (n={ok.sum()})")
        v = T[lo:hi, 14]
        print(f"   {'exit':>16s}: min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us")
    # per-CU occupancy: how many workgroups per CU, and whether the two of a CU overlap
    from collections import defaultdict
    per = defaultdict(list)
    for i in range(len(wg)):
        per[int(cukey[i])].append(i)
    cnt = defaultdict(int)
    for k, v in per.items(): cnt[len(v)] += 1
    print("workgroups per CU histogram:", dict(cnt))
    np.save(os.path.join(REPO, "gpurun_out", "wg_timeline.npy"), wg)
    for w in range(8):
        s = st[w][:13]
        if s[0] == 0: continue
        idx = [i for i in range(13) if s[i]]
        print(f"wave {w}: " + " ".join(f"{a}>{b}:{int(s[b] - s[a]):>6d}" for a, b in zip(idx[:-1], idx[1:])) + f"   total {int(s[12]-s[0])}")
