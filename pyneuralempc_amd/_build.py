"""Build libnempc.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache, no torch extension:
the product boundary is a plain C ABI (include/nempc.h) loaded with ctypes.

Every translation unit goes through the stages hipcc runs itself, with one more in the middle:

    hipcc --cuda-device-only -S        device code as gfx950 assembly
    _isa.repair / _isa.scan            vector instructions the compiler left in front of an exec restore are moved behind
                                       it (see _isa.py); the repaired text must scan clean or the build fails
    clang (assembler) + lld            code object
    clang-offload-bundler              fat binary
    hipcc --cuda-host-only             host object with the fat binary embedded (-fcuda-include-gpubinary)

The report of the middle stage stays next to the object (`build/<unit>.isa.json`).
"""
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

from . import _isa

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnempc.so")
# the matrix-core kernels: one translation unit per (dtype, hidden activation); tanh (the headline) first, it is the longest
_MFMA_ACTS = ["relu", "sigmoid", "softplus", "elu"]          # (+ "rt": per-layer codes at run time)
SOURCES = (["kernels_mfma_f64_rt.hip", "kernels_mfma_f64.hip", "kernels_mfma_f32_rt.hip", "kernels_mfma_f32.hip", "solver.hip"] +
           [f"kernels_mfma_{t}_{a}.hip" for t in ("f64", "f32") for a in _MFMA_ACTS] +
           ["nempc_api.hip", "kernels_valu.hip", "kernels_layered.hip", "kernels_post.hip", "kernels_mfma.hip", "kernels_rk4hess.hip",
            "comm.hip"])
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


# per-source extra flags.  Kernel-argument preload: the leading 16 dwords of a kernel's plain arguments are placed in
# scalar registers by the dispatcher (the fixed-shape row kernel lists what its first loads need there)
EXTRA_FLAGS = {f: ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
               for f in ["kernels_mfma_f64.hip", "kernels_mfma_f64_rt.hip"] + [f"kernels_mfma_f64_{a}.hip" for a in _MFMA_ACTS]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnempc.so cannot be built")
    return exe


def _llvm_tool(name):
    """clang / lld / clang-offload-bundler of the ROCm installation hipcc belongs to."""
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(_hipcc())))
    for d in (os.path.join(rocm, "lib", "llvm", "bin"), "/opt/rocm/lib/llvm/bin"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    raise RuntimeError(f"{name} not found next to hipcc: libnempc.so cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, what):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{what} failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")


def compile_unit(src, obj, extra=(), keep_asm=False, repair=True):
    """One translation unit through the staged pipeline (module docstring).  -> the middle stage's report.
    repair=False (tools/build_variant.py --no-repair: the control of an A/B) assembles the compiler's text as it is."""
    stem = obj[:-2] if obj.endswith(".o") else obj
    asm, dev, hsaco, fb = stem + ".s", stem + ".dev.o", stem + ".hsaco", stem + ".hipfb"
    flags = FLAGS + list(extra)
    _run([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", asm, src], f"hipcc -S {src}")
    with open(asm) as fh:
        text = fh.read()
    text, found = _isa.repair(text) if repair else (text, [])
    left = _isa.scan(text)
    hazards = _isa.scan_store_hazard(text)
    report = {"unit": os.path.basename(src), "repaired": found, "left": left, "store_hazards": hazards}
    with open(stem + ".isa.json", "w") as fh:
        json.dump(report, fh, indent=1)
    if left and repair:
        raise RuntimeError(f"{src}: vector instructions in front of an exec restore survived the repair: {left}")
    if hazards and repair:
        raise RuntimeError(f"{src}: an inline-asm store of > 64 bits has its data registers overwritten inside the hazard "
                           f"window (put `s_nop 1` behind the store in the asm statement): {hazards}")
    if found:
        with open(asm, "w") as fh:
            fh.write(text)
    _run([_llvm_tool("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", asm, "-o", dev],
         f"assembling {asm}")
    _run([_llvm_tool("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev],
         f"linking {hsaco}")
    _run([_llvm_tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={hsaco}",
          f"-output={fb}"], f"bundling {fb}")
    _run([_hipcc()] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", obj],
         f"hipcc host {src}")
    for tmp in ([dev, hsaco, fb] + ([] if keep_asm else [asm])):
        if os.path.exists(tmp):
            os.remove(tmp)
    return report


def build(force=False, verbose=True):
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(REPO, "include", "nempc.h"))
    headers += [os.path.join(PKG, "_isa.py"), os.path.join(PKG, "_build.py")]      # the pipeline is a dependency as well
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers) or not os.path.exists(o[:-2] + ".isa.json"):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        return s, compile_unit(s, o, EXTRA_FLAGS.get(os.path.basename(s), []))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, rep in ex.map(compile_one, jobs):
                if verbose:
                    n = len(rep["repaired"])
                    note = f" ({n} exec-restore block{'s' if n != 1 else ''} repaired)" if n else ""
                    print(f"[nempc build] compiled {os.path.basename(s)}{note}", file=sys.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        _run(cmd, "link")
        if verbose:
            print(f"[nempc build] linked {LIB}", file=sys.stderr)
    return LIB


def isa_reports():
    """The middle stage's reports of the last build, one per translation unit."""
    objdir = os.path.join(PKG, "build")
    out = []
    for src in SOURCES:
        p = os.path.join(objdir, src.replace(".hip", ".isa.json"))
        if os.path.exists(p):
            with open(p) as fh:
                out.append(json.load(fh))
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
