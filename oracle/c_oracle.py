"""ctypes wrapper of oracle/_build/liboracle.so (C/OpenMP oracle) -- TEST INFRASTRUCTURE ONLY.
Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
MAXL = 8


class _CProblem(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int), ("nx", ctypes.c_int), ("nu", ctypes.c_int), ("kind", ctypes.c_int),
                ("DT", ctypes.c_double), ("nl", ctypes.c_int), ("din", ctypes.c_int * MAXL),
                ("dout", ctypes.c_int * MAXL), ("W", ctypes.c_void_p * MAXL), ("b", ctypes.c_void_p * MAXL),
                ("Q", ctypes.c_void_p), ("R", ctypes.c_void_p), ("xref", ctypes.c_void_p), ("uref", ctypes.c_void_p),
                ("cx", ctypes.c_void_p), ("cu", ctypes.c_void_p), ("box", ctypes.c_int), ("act", ctypes.c_int * MAXL),
                ("actp", ctypes.c_double * MAXL)]


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "nempc_oracle.c")):
        subprocess.run(["make", "-C", HERE, "-B"], check=True, capture_output=True)
    return LIB


class COracle:
    def __init__(self, prob):
        """prob: oracle.nempc_oracle.Problem"""
        if not os.path.exists(LIB):
            build()
        self.lib = ctypes.CDLL(LIB)
        self.lib.oracle_eval.restype = ctypes.c_int
        self.prob = prob
        self._keep = []
        c = _CProblem()
        c.H, c.nx, c.nu, c.kind, c.DT = prob.H, prob.nx, prob.nu, prob.kind, prob.DT
        c.nl = len(prob.net.W)
        for l, (w, b) in enumerate(zip(prob.net.W, prob.net.b)):
            w, b = np.ascontiguousarray(w), np.ascontiguousarray(b)
            self._keep += [w, b]
            c.din[l], c.dout[l] = w.shape
            c.W[l], c.b[l] = w.ctypes.data, b.ctypes.data
        for name in ("Q", "R", "xref", "uref", "cx", "cu"):
            a = np.ascontiguousarray(getattr(prob, name), dtype=np.float64)
            self._keep.append(a)
            setattr(c, name, a.ctypes.data)
        c.box = int(prob.box is not None)
        from .nempc_oracle import ACT_IDS, act_split
        for l, spec in enumerate(prob.net.act):
            name, par = act_split(spec)
            c.act[l], c.actp[l] = ACT_IDS[name], par
        self.c = c

    def eval(self, Z, X0, dense=True, nthreads=0):
        Z, X0 = np.ascontiguousarray(Z, dtype=np.float64), np.ascontiguousarray(X0, dtype=np.float64)
        B, p = Z.shape[0], self.prob
        f, grad, g = np.empty(B), np.empty((B, p.n)), np.empty((B, p.m))
        jac = np.empty((B, p.m, p.n)) if dense else None
        vp = ctypes.c_void_p
        used = self.lib.oracle_eval(ctypes.byref(self.c), ctypes.c_int(B), vp(Z.ctypes.data), vp(X0.ctypes.data),
                                    vp(f.ctypes.data), vp(grad.ctypes.data), vp(g.ctypes.data),
                                    vp(jac.ctypes.data) if dense else None, ctypes.c_int(nthreads))
        self.threads_used = used
        return f, grad, g, jac
