#!/usr/bin/env python3
"""Per-workgroup timeline of ONE product of the layer-at-a-time path (diagnostic build, never shipped): builds
pyneuralempc_amd/build_stamps_lg/libnempc_stamps_lg.so with -DNEMPC_STAMPS (kernels_layered.hip: LG_WGSTAMP), runs a 2/1,
2 x 256 tanh, H = 20, B = 1024 evaluation in fp64 and prints where the workgroups of the chosen product spend their time.
   python tools/diag_stamps_layered.py [--build-only]      NEMPC_LG_STAMP = 10 SEED + CONTRACT (11: the reverse product with
   seed loader and J contraction, 2: the last hidden layer with the output contraction, 0: a plain product)"""
import ctypes, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyneuralempc_amd import _build, _lib


def build():
    out = os.path.join(_build.PKG, "build_stamps_lg")
    os.makedirs(out, exist_ok=True)
    _build.build(verbose=False)
    stamped = {"nempc_api.hip", "kernels_layered.hip"}
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
    lib = os.path.join(out, "libnempc_stamps_lg.so")
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
    return lib


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        print(build()); sys.exit(0)
    import torch
    from oracle import nempc_oracle as orc
    _lib.LIB_PATH = os.path.join(_build.PKG, "build_stamps_lg", "libnempc_stamps_lg.so")
    _lib._lib = None
    from pyneuralempc_amd import CallbackEngine
    B, H, nx, nu = 1024, 20, 2, 1
    hidden = [int(x) for x in os.environ.get("HIDDEN", "256,256").split(",")]
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel="layered")
    eng.lib.nempc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.nempc_debug_stamps(eng._handle, None)
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    Z, X0 = eng.to_device(Z), eng.to_device(X0)
    for _ in range(3):
        eng.eval(Z, X0, ("g", "jac_tiles"))
    buf = np.zeros(1024 + 4096 * 16, dtype=np.int64)
    eng.lib.nempc_debug_stamps(eng._handle, buf.ctypes.data_as(ctypes.c_void_p))
    wg = buf[1024:].reshape(4096, 16)
    wg = wg[wg[:, 0] != 0]
    f_mhz = float(os.environ.get("NEMPC_CLK_MHZ", "2200"))
    real_exit_us = wg[:, 13] / 100.0
    T = np.zeros(wg.shape)
    for i in range(15):
        T[:, i] = real_exit_us - (wg[:, 14] - wg[:, i]) / f_mhz
    T -= T[:, 0].min()
    T[wg == 0] = np.nan
    hw = wg[:, 15]
    xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
    cukey = xcc * 1000 + ((hwid >> 13) & 7) * 100 + ((hwid >> 12) & 1) * 10 + ((hwid >> 8) & 0xF)
    print(f"product NEMPC_LG_STAMP={os.environ.get('NEMPC_LG_STAMP', '11')}: workgroups {len(wg)}, distinct CUs {len(set(cukey.tolist()))}, "
          f"kernel span {np.nanmax(T[:, 14]):.1f} us (shader clock assumed {f_mhz:.0f} MHz)")
    names = {0: "entry", 1: "first two chunks asked for", 2: "first barrier", 3: "main loop done", 4: "epilogue operands in hand", 14: "exit"}
    seg = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 14), (0, 14)]
    for a, b in seg:
        d = T[:, b] - T[:, a]
        d = d[~np.isnan(d)]
        if len(d):
            print(f"   {names[a]:>28s} -> {names[b]:<28s}: min {d.min():6.2f}  p10 {np.percentile(d, 10):6.2f}  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    # wave 0's time inside the main loop by segment (sums over the chunks)
    seg = wg[:, 5:9].astype(np.float64) / f_mhz
    tot = seg.sum(axis=1)
    print("   main loop of wave 0, us per workgroup (median / p10 / p90): " + "   ".join(
        f"{nm} {np.median(seg[:, i]):.2f} / {np.percentile(seg[:, i], 10):.2f} / {np.percentile(seg[:, i], 90):.2f}"
        for i, nm in enumerate(["issue loads", "LDS reads + matrix instr.", "wait loads + LDS writes", "barrier"])))
    late = T[:, 0] > 0.6 * np.nanmax(T[:, 14])          # the last round: two workgroups per CU
    for label, sel in (("first rounds", ~late), ("last round ", late)):
        if sel.any():
            print(f"      {label}: " + "   ".join(f"{np.median(seg[sel, i]):.2f}" for i in range(4)) + f"   (n = {int(sel.sum())})")
    # per CU: how many workgroups are in their main loop (matrix instructions) at a time
    span = np.nanmax(T[:, 14])
    grid = np.linspace(0, span, 400)
    inloop = np.zeros((len(grid),)); resident = np.zeros((len(grid),)); ncu = len(set(cukey.tolist()))
    for i in range(len(wg)):
        inloop += (grid >= T[i, 2]) & (grid < T[i, 3])
        resident += (grid >= T[i, 0]) & (grid < T[i, 14])
    inloop /= ncu; resident /= ncu
    print("   time (us)      resident WG / CU    in main loop / CU")
    for k in range(0, len(grid), 20):
        print(f"   {grid[k]:8.1f}      {resident[k]:6.2f}              {inloop[k]:6.2f}")
    # a CU's own schedule: entries / exits of its workgroups
    one = sorted(set(cukey.tolist()))[0]
    idx = [i for i in range(len(wg)) if cukey[i] == one]
    idx.sort(key=lambda i: T[i, 0])
    print(f"   workgroups of CU {one}: entry / first barrier / loop done / operands / exit (us)")
    for i in idx:
        print("      " + "  ".join(f"{T[i, k]:7.2f}" for k in (0, 2, 3, 4, 14)))
    # fraction of CU-time with 0, 1, 2, 3, 4+ workgroups in their main loop
    hist = np.zeros(6)
    for cu in set(cukey.tolist()):
        ii = [i for i in range(len(wg)) if cukey[i] == cu]
        cnt = np.zeros(len(grid))
        for i in ii:
            cnt += (grid >= T[i, 2]) & (grid < T[i, 3])
        for v in cnt:
            hist[min(int(v), 5)] += 1
    hist /= hist.sum()
    print("   share of CU-time with k workgroups in their main loop, k = 0..5+: " + " ".join(f"{v:.3f}" for v in hist))
