"""Build libnempc.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache, no torch extension:
the product boundary is a plain C ABI (include/nempc.h) loaded with ctypes."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnempc.so")
# the matrix-core kernels: one translation unit per (dtype, hidden activation); tanh (the headline) first, it is the longest
_MFMA_ACTS = ["relu", "sigmoid", "softplus", "elu"]
SOURCES = (["kernels_mfma_f64.hip", "kernels_mfma_f32.hip", "solver.hip"] +
           [f"kernels_mfma_{t}_{a}.hip" for t in ("f64", "f32") for a in _MFMA_ACTS] +
           ["nempc_api.hip", "kernels_valu.hip", "kernels_post.hip", "kernels_mfma.hip", "kernels_rk4hess.hip", "comm.hip"])
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


# per-source extra flags.  Kernel-argument preload: the leading 16 dwords of a kernel's plain arguments are placed in
# scalar registers by the dispatcher (the fixed-shape row kernel lists what its first loads need there)
EXTRA_FLAGS = {f: ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
               for f in ["kernels_mfma_f64.hip"] + [f"kernels_mfma_f64_{a}.hip" for a in _MFMA_ACTS]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnempc.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(REPO, "include", "nempc.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        return s

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[nempc build] compiled {os.path.basename(s)}", file=sys.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[nempc build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
