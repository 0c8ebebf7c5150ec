"""Solver-glue interfaces (reference: pyNeuralEMPC/optimizer/base.py:7-149).

The object handed to cyipopt / scipy is a ProblemInterface; its five callbacks are the drop-in
boundary of this build: behind them sits one fused device evaluation per iterate (``_CallbackGlue``).
"""
import numpy as np
import torch

from ..constraints import BoxStateConstraint, Constraint, DomainConstraint  # noqa: F401
from ..integrator.base import DeviceIntegrator
from ..objective.quadratic import QuadraticObjective


class ProblemInterface:
    def __init__(self, use_hessian: bool):
        self.use_hessian = use_hessian

    def objective(self, x):
        raise NotImplementedError("")

    def gradient(self, x):
        raise NotImplementedError("")

    def constraints(self, x):
        raise NotImplementedError("")

    def hessianstructure(self):
        raise NotImplementedError("")

    def hessian(self, x, lagrange, obj_factor):
        raise NotImplementedError("")

    def jacobian(self, x):
        raise NotImplementedError("")

    def get_constraint_lower_bounds(self):
        raise NotImplementedError("")

    def get_constraint_upper_bounds(self):
        raise NotImplementedError("")

    def get_init_value(self):
        raise NotImplementedError("")

    def get_init_variables(self):
        raise NotImplementedError("")


class ProblemInterfaceHessianFree:
    """View of a problem without hessian / hessianstructure, so that Ipopt falls back to its
    quasi-Newton update (reference: optimizer/base.py:7-32)."""

    def __init__(self, core):
        self.core = core
        for name in ("objective", "gradient", "constraints", "jacobian", "get_constraint_lower_bounds",
                     "get_constraint_upper_bounds", "get_init_value"):
            setattr(self, name, getattr(core, name))


class ProblemFactory:
    def __init__(self):
        self.x0 = None
        self.p = None
        self.tvp = None
        self.objective = None
        self.constraints = None
        self.use_hessian = False
        self.integrator = None
        self.init_u, self.init_x = None, None

    def getProblemInterface(self) -> ProblemInterface:
        for value, name in ((self.x0, "x0"), (self.objective, "objective"), (self.constraints, "constraints"),
                            (self.integrator, "integrator")):
            if value is None:
                raise RuntimeError(f"Not ready yet ! {name} is missing")
        return self._process()

    def set_integrator(self, integrator):
        self.integrator = integrator

    def set_x0(self, x0):
        self.x0 = x0

    def set_init_values(self, init_x, init_u):
        self.init_x = init_x
        self.init_u = init_u

    def set_p(self, p):
        self.p = p

    def set_tvp(self, tvp):
        self.tvp = tvp

    def set_objective(self, obj):
        self.objective = obj

    def set_constraints(self, ctrs: list):
        self.constraints = ctrs

    def set_use_hessian(self, hessian: bool):
        self.use_hessian = hessian

    def _process(self):
        raise NotImplementedError("")


class Optimizer:
    FAIL = 1
    SUCCESS = 0

    def __init__(self):
        pass

    def get_factory(self) -> ProblemFactory:
        """Return the solver's associated factory."""

    def solve(self, problem: ProblemInterface, domain_constraint: DomainConstraint):
        raise NotImplementedError("")


def cold_start(x0, H, u_dim):
    """[x0 tiled H ; zeros(H*u_dim)]  (reference: optimizer/ipopt.py:149, slsqp.py:163)."""
    return np.concatenate([np.tile(np.asarray(x0, dtype=np.float64), H), np.zeros(H * u_dim)])


def warm_start_shift(prev, H, x_dim, u_dim):
    """Previous solution advanced by one step, last step repeated (ipopt.py:141-147, slsqp.py:155-161)."""
    prev = np.asarray(prev, dtype=np.float64)
    xs, us = prev[:H * x_dim].reshape(H, x_dim), prev[H * x_dim:].reshape(H, u_dim)
    return np.concatenate([xs[1:].ravel(), xs[-1], us[1:].ravel(), us[-1]])


_FUSED_CACHE_SIZE = 4   # evaluators an integrator keeps alive (each owns a device handle and its workspaces)


def fused_evaluator(integrator, objective, box):
    """The integrator's evaluator for (objective, box), created on first use and REFRESHED on every use: the cache is
    keyed on the objects themselves (weak references, so a new objective allocated at a recycled id() cannot hit a stale
    entry) and the uploaded parameters are compared with the objects' current ones, because the caller may edit
    QuadraticObjective.params / the box bounds between solves and the reference re-reads them on every call.  At most
    _FUSED_CACHE_SIZE entries; the least recently used one is dropped (with its device handle)."""
    import weakref
    cache = integrator._fused
    key = (id(objective), id(box) if box is not None else None)
    ent = cache.get(key)
    if ent is not None and (ent[0]() is not objective or (box is not None and ent[1]() is not box)):
        del cache[key]          # a dead object's id was recycled
        ent = None
    if ent is None:
        ev = _FusedEvaluator(integrator, objective, box)
        ent = (weakref.ref(objective), weakref.ref(box) if box is not None else None, ev)
    else:
        del cache[key]          # re-inserted below: dict order = recency
    cache[key] = ent
    while len(cache) > _FUSED_CACHE_SIZE:
        cache.pop(next(iter(cache)))
    ent[2].refresh(objective, box)
    return ent[2]


class _CoherentHostBuffer:
    """Host memory the device addresses directly AND that is coherent with the host while a kernel runs: hipHostMalloc with
    hipHostMallocCoherent | hipHostMallocMapped, asked for explicitly (what a default hipHostMalloc -- torch's
    pin_memory=True -- gives depends on HIP_HOST_COHERENT).  Fine-grained pages are mapped uncached on the device: a kernel's
    stores are not held in the GPU's L2 for an end-of-kernel write-back, they leave for host memory as PCIe posted writes when
    they are issued.  `array(dtype)` is a NumPy view of the whole buffer; `ptr` is valid on host and device."""

    _hip = None
    COHERENT, MAPPED = 0x40000000, 0x2          # hipHostMallocCoherent, hipHostMallocMapped (hip_runtime_api.h)

    def __init__(self, nbytes):
        import ctypes
        if _CoherentHostBuffer._hip is None:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
            hip.hipHostMalloc.restype = ctypes.c_int
            hip.hipHostFree.argtypes = [ctypes.c_void_p]
            hip.hipHostFree.restype = ctypes.c_int
            _CoherentHostBuffer._hip = hip
        self.nbytes = max(int(nbytes), 64)
        p = ctypes.c_void_p()
        rc = _CoherentHostBuffer._hip.hipHostMalloc(ctypes.byref(p), self.nbytes, self.COHERENT | self.MAPPED)
        if rc != 0 or not p.value:
            raise RuntimeError(f"hipHostMalloc(coherent, {self.nbytes} bytes) failed with HIP error {rc}")
        self.ptr = p.value
        self._raw = (ctypes.c_ubyte * self.nbytes).from_address(self.ptr)
        self.array(np.uint8)[:] = 0

    def array(self, dtype):
        return np.frombuffer(self._raw, dtype=dtype)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._raw = None
                _CoherentHostBuffer._hip.hipHostFree(self.ptr)
                self.ptr = None
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


class _StreamWaiter:
    """Wait for a stream's work from the host without a stream synchronisation: the stream writes a sequence number into a
    word of coherent host memory after the queued kernel (hipStreamWriteValue32), the host spins on that word.  2.9 us less
    per B = 1 callback than stream.synchronize() (26.8 -> 24.0 us, tools/host_sync_probe.py).

    Why the kernel's results are visible when the word is -- the contract this relies on, piece by piece:
    (1) the result buffer and the word are `_CoherentHostBuffer`s: fine-grained host memory, uncached on the device, so the
        kernel's stores do not wait in L2 for the end-of-kernel release; they travel to host memory as posted writes in the
        order the memory system issues them and are complete before the kernel's completion is signalled (the dispatch
        packet's release fence is system scope for fine-grained memory);
    (2) hipStreamWriteValue32 is stream-ordered: "the write is performed after all earlier commands on the stream have
        completed" (HIP API); it is a packet of its own behind the kernel's, with the barrier bit, so its write is issued
        after (1)'s writes have been acknowledged;
    (3) the host reads the word, THEN the results (x86 loads are not reordered with older loads; NumPy copies out after the
        spin returns).
    `tests/test_gpu_plugins.py::test_waiter_path_matches_the_synchronised_path_bit_for_bit` is the backstop: a poisoned result
    buffer before every call, alternating inputs, a few hundred callbacks against stream.synchronize().
    A spin that does not end within a second falls back to the synchronisation, which surfaces a device fault as an error; a
    HIP runtime without the entry point uses the synchronisation from the start."""

    _fn = None
    _probed = False
    SPIN_SECONDS = 1.0

    def __init__(self, stream):
        import ctypes
        self.stream = stream
        self.seq = 0
        self.flag = None
        if not _StreamWaiter._probed:
            _StreamWaiter._probed = True
            try:
                hip = ctypes.CDLL("libamdhip64.so")
                fn = hip.hipStreamWriteValue32
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
                fn.restype = ctypes.c_int
                _StreamWaiter._fn = fn
            except (OSError, AttributeError):
                _StreamWaiter._fn = None
        if _StreamWaiter._fn is not None:
            self.flag = _CoherentHostBuffer(64)
            self.word = self.flag.array(np.int32)
            self.fptr = ctypes.c_void_p(self.flag.ptr)
            self.sptr = ctypes.c_void_p(stream.cuda_stream)

    def wait(self):
        if self.flag is None:
            self.stream.synchronize()
            return
        self.seq = (self.seq % 0x7fffffff) + 1
        if _StreamWaiter._fn(self.sptr, self.fptr, self.seq, 0) != 0:
            self.flag = None                    # not supported on this stream / memory: synchronise from now on
            self.stream.synchronize()
            return
        word, seq = self.word, self.seq
        import time
        deadline = None
        while True:
            for _ in range(2048):
                if word[0] == seq:
                    return
            now = time.perf_counter()
            if deadline is None:
                deadline = now + self.SPIN_SECONDS
            elif now > deadline:
                break
        self.stream.synchronize()               # (a device fault ends here as an exception instead of an endless spin)


class _FusedEvaluator:
    """integrator + QuadraticObjective [+ BoxStateConstraint] on ONE engine: a single device call
    yields f, grad f, g, dense jac g for an iterate; results are cached per (z, x0) because the
    solvers ask for the four pieces in separate callbacks."""

    def __init__(self, integrator, objective, box):
        model = integrator.model
        self.model = model
        self.H, self.nx, self.nu = integrator.H, model.x_dim, model.u_dim
        self.engine = model.make_engine(integrator.H, integrator.KIND, DT=integrator.DT, max_batch=1)
        self._obj_fp = None
        self._box_fp = None
        self._fast_state = {}
        self._hess_state = None
        self._key = None
        self._val = None
        self.n_device_evals = 0
        self.refresh(objective, box)

    def refresh(self, objective, box):
        """Re-upload the objective / box bounds when they differ from what the handle holds."""
        fp = objective.fingerprint(self.H, self.nx, self.nu)
        if fp != self._obj_fp:
            self.engine.set_objective(**objective.resolved(self.H, self.nx, self.nu))
            self._obj_fp = fp
            self._key = None
        if box is not None:
            lo, hi = box._bounds(self.nx)
            bfp = np.asarray(lo, dtype=np.float64).tobytes() + np.asarray(hi, dtype=np.float64).tobytes()
            if bfp != self._box_fp:
                self.engine.set_box_rows(lo, hi)
                self._box_fp = bfp
                self._key = None

    def set_parameters(self, p, tvp):
        """Bind the problem's constant / time-varying parameters (extra network inputs; the history
        of a rolling-window model) for the next evaluations."""
        if self.model.bind_inputs(self.engine, p, tvp):
            self._key = None

    # ---- B = 1 hot path (one NLP iterate of Ipopt / SLSQP): no per-call allocation, no explicit copies.
    # Inputs and outputs live in two PINNED host buffers that the device addresses directly (hipHostMalloc memory is
    # mapped into the GPU's address space): the kernels read z / x0 over PCIe (a few hundred bytes) and stream their
    # results straight into host memory, so a callback costs one ctypes call, the launch and one stream wait instead of
    # two H2D copies, the launches and four D2H copies (115 us -> ~30 us per evaluation, tools/dropin_latency.py).
    def _fast(self, sparse):
        import ctypes
        eng = self.engine
        key = (eng.n, eng.m, eng.nnz_jac, bool(sparse), eng._handle.value)
        st = self._fast_state.get(bool(sparse))
        if st is not None and st["key"] == key:
            return st
        n, m, nx = eng.n, eng.m, eng.nx
        nj = eng.nnz_jac if sparse else m * n
        npdt = np.float64 if eng.dtype == torch.float64 else np.float32
        esz = np.dtype(npdt).itemsize
        hin = _CoherentHostBuffer((n + nx) * esz)
        hout = _CoherentHostBuffer((1 + n + m + nj) * esz)
        vp = ctypes.c_void_p
        stream = torch.cuda.Stream(eng.device)
        outp = hout.ptr
        st = dict(key=key, hin=hin, hout=hout, zin=hin.array(npdt)[:n], xin=hin.array(npdt)[n:n + nx],
                  out=hout.array(npdt)[:1 + n + m + nj], stream=stream,
                  waiter=_StreamWaiter(stream),
                  n=n, m=m, nj=nj, sparse=bool(sparse),
                  args=(eng._handle, 1, vp(hin.ptr), vp(hin.ptr + n * esz), vp(outp), vp(outp + esz),
                        vp(outp + (1 + n) * esz), None if sparse else vp(outp + (1 + n + m) * esz), None,
                        vp(outp + (1 + n + m) * esz) if sparse else None, vp(stream.cuda_stream)))
        self._fast_state[bool(sparse)] = st
        return st

    def evaluate(self, z, x0, sparse=False):
        """f, grad, g and the Jacobian (dense (m,n), or the band values in jac_structure() order with sparse=True) at
        one iterate; cached per (z, x0) because the solvers ask for the pieces in separate callbacks."""
        z = np.asarray(z, dtype=np.float64)
        if (self._key is not None and self._key[2] == bool(sparse) and np.array_equal(self._key[0], z)
                and np.array_equal(self._key[1], x0)):
            return self._val
        from .. import _lib
        st = self._fast(sparse)
        self.engine._check_extra(1)
        st["zin"][:] = z
        st["xin"][:] = x0
        rc = self.engine.lib.nempc_eval(*st["args"])      # (the library selects the handle's device itself)
        if rc:
            _lib.check(rc)
        st["waiter"].wait()
        o, n, m, nj = st["out"], st["n"], st["m"], st["nj"]
        jac = o[1 + n + m:1 + n + m + nj].astype(np.float64)
        self._val = {"f": float(o[0]), "grad": o[1:1 + n].astype(np.float64), "g": o[1 + n:1 + n + m].astype(np.float64),
                     ("jac_sparse" if sparse else "jac_dense"): jac if sparse else jac.reshape(m, n)}
        self._key = (z.copy(), np.array(x0, dtype=np.float64), bool(sparse))
        self.n_device_evals += 1
        return self._val

    def hessian_values(self, z, x0, lagrange, obj_factor):
        import ctypes
        eng = self.engine
        hs = self._hess_state
        key = (eng.n, eng.m, eng.nnz_hess, eng._handle.value)
        if hs is None or hs["key"] != key:
            n, m, nx, nh = eng.n, eng.m, eng.nx, eng.nnz_hess
            npdt = np.float64 if eng.dtype == torch.float64 else np.float32
            esz = np.dtype(npdt).itemsize
            hin = _CoherentHostBuffer((n + nx + m + 1) * esz)
            hout = _CoherentHostBuffer(nh * esz)
            vp = ctypes.c_void_p
            stream = torch.cuda.Stream(eng.device)
            base = hin.ptr
            hs = dict(key=key, hin=hin, hout=hout, inp=hin.array(npdt)[:n + nx + m + 1], out=hout.array(npdt)[:nh], stream=stream,
                      waiter=_StreamWaiter(stream),
                      args=(eng._handle, 1, vp(base), vp(base + n * esz), vp(base + (n + nx) * esz),
                            vp(base + (n + nx + m) * esz), vp(hout.ptr), None, None, vp(stream.cuda_stream)))
            self._hess_state = hs
        from .. import _lib
        n, m, nx = eng.n, eng.m, eng.nx
        eng._check_extra(1)
        hs["inp"][:n] = z
        hs["inp"][n:n + nx] = x0
        hs["inp"][n + nx:n + nx + m] = lagrange
        hs["inp"][n + nx + m] = obj_factor
        rc = eng.lib.nempc_hess(*hs["args"])
        if rc:
            _lib.check(rc)
        hs["waiter"].wait()
        return hs["out"].astype(np.float64)


class _CallbackGlue:
    """Shared body of IpoptProblem / SlsqpProblem: split z, fan out, concatenate
    (reference: optimizer/ipopt.py:20-108, slsqp.py:25-110)."""

    def _setup(self, x0, objective_func, constraints, integrator, p, tvp):
        self.x0 = x0
        self.objective_func = objective_func
        self.constraints_list = constraints
        self.integrator = integrator
        model = integrator.model
        self.x_dim, self.u_dim, self.p_dim, self.tvp_dim = model.x_dim, model.u_dim, model.p_dim, model.tvp_dim
        self.H = integrator.H
        self.p = p
        self.tvp = tvp
        self._fused = None
        boxes = [c for c in constraints if isinstance(c, BoxStateConstraint)]
        if (isinstance(integrator, DeviceIntegrator) and integrator.on_device and isinstance(objective_func, QuadraticObjective)
                and len(boxes) == len(constraints) and len(boxes) <= 1):
            self._fused = fused_evaluator(integrator, objective_func, boxes[0] if boxes else None)
            self._fused.set_parameters(p, tvp)

    def _split(self, x):
        nxh = self.x_dim * self.H
        states = x[:nxh].reshape(self.H, self.x_dim)
        u = x[nxh:nxh + self.u_dim * self.H].reshape(self.H, self.u_dim)
        return states, u, self.tvp, self.p

    def _objective(self, x):
        if self._fused is not None:
            return float(self._fused.evaluate(x, self.x0)["f"])
        states, u, tvp, p = self._split(x)
        return self.objective_func.forward(states, u, p=p, tvp=tvp)

    def _gradient(self, x):
        if self._fused is not None:
            return self._fused.evaluate(x, self.x0)["grad"].copy()
        states, u, tvp, p = self._split(x)
        return self.objective_func.gradient(states, u, p=p, tvp=tvp)

    def _all_constraints(self, x):
        if self._fused is not None:
            return self._fused.evaluate(x, self.x0)["g"].copy()
        states, u, tvp, p = self._split(x)
        parts = [self.integrator.forward(states, u, self.x0, p=p, tvp=tvp)]
        parts += [c.forward(states, u, p=p, tvp=tvp) for c in self.constraints_list]
        return np.concatenate(parts)

    def _all_jacobian(self, x):
        if self._fused is not None:
            return self._fused.evaluate(x, self.x0)["jac_dense"].copy()
        states, u, tvp, p = self._split(x)
        parts = [self.integrator.jacobian(states, u, self.x0, p=p, tvp=tvp)]
        parts += [c.jacobian(states, u, p=p, tvp=tvp) for c in self.constraints_list]
        return np.concatenate(parts, axis=0)

    def get_constraint_lower_bounds(self):
        return np.concatenate([np.asarray(c.get_lower_bounds(self.H), dtype=np.float64)
                               for c in [self.integrator, ] + self.constraints_list])

    def get_constraint_upper_bounds(self):
        return np.concatenate([np.asarray(c.get_upper_bounds(self.H), dtype=np.float64)
                               for c in [self.integrator, ] + self.constraints_list])

    def get_init_value(self):
        return self.x0

    def get_init_variables(self):
        return self.init_x, self.init_u
