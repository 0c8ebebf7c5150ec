#!/usr/bin/env python3
"""Build a variant of libnempc.so with extra -D flags into pyneuralempc_amd/build_<tag>/libnempc_<tag>.so (A/B kernel
experiments; run with NEMPC_LIB=<that path>).   python tools/build_variant.py <tag> [-DNAME=VALUE ...] [--only a.hip,b.hip]
--no-repair: assemble the compiler's text as it is (the control for the exec-restore repair of _isa.py)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyneuralempc_amd import _build

def main():
    tag = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-D") or a in ("-O1", "-O2", "-O0")]
    for a in sys.argv[2:]:          # --mllvm=<option>: passed to the backend as -mllvm <option> (scheduling experiments)
        if a.startswith("--mllvm="):
            defs += ["-mllvm", a[len("--mllvm="):]]
    out = os.path.join(_build.PKG, f"build_{tag}")
    os.makedirs(out, exist_ok=True)
    # only the translation units that see the kernels are rebuilt with the flags; the rest come from the main build
    kern = {"kernels_mfma_f64.hip", "kernels_mfma_f32.hip"} if "--all" not in sys.argv else set(_build.SOURCES)
    if "--only" in sys.argv:
        kern = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    _build.build(verbose=False)
    def one(src):
        if src in kern:
            o = os.path.join(out, src.replace(".hip", ".o"))
            extra = (_build.EXTRA_FLAGS.get(src, []) if "--no-extra" not in sys.argv else []) + defs
            _build.compile_unit(os.path.join(_build.CSRC, src), o, extra, repair="--no-repair" not in sys.argv)
            return o
        return os.path.join(_build.PKG, "build", src.replace(".hip", ".o"))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, _build.SOURCES))
    lib = os.path.join(out, f"libnempc_{tag}.so")
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
    print(lib)

if __name__ == "__main__":
    main()
