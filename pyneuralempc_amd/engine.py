"""Batched callback engine: the thin host wrapper over the C ABI.

One ``CallbackEngine`` = one ``nempc_handle``: a fixed problem family (dims, integrator, network,
objective, optional box rows) evaluated for a batch of B independent problems per call on one
MI355X.  torch is used for device memory and streams only; every number is produced by the HIP
kernels in csrc/.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_TORCH_DTYPES = {torch.float64: _lib.F64, torch.float32: _lib.F32}


def resolve_activations(activations, n_layers):
    """Per-layer activation names of an n_layers-deep dense stack.  None: tanh hidden layers and a linear output layer
    (the reference's nn_model.h5); one name: that activation on every hidden layer, linear output; a sequence: one name
    per layer, the output layer included.  Names as Keras spells them (_lib.ACTIVATION_IDS); "elu:0.5" / "leaky_relu:0.1"
    carry their alpha (_lib.split_activation)."""
    if activations is None:
        names = ["tanh"] * (n_layers - 1) + ["linear"]
    elif isinstance(activations, str):
        names = [activations] * (n_layers - 1) + ["linear"]
    else:
        names = [str(a) for a in activations]
        if len(names) != n_layers:
            raise ValueError(f"activations: expected one name per dense layer ({n_layers}), got {len(names)}")
    for a in names:
        _lib.split_activation(a)
    return names


def _as_c_double(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class CallbackEngine:
    def __init__(self, weights, biases, H, nx, nu, integrator="discret", DT=1.0, dtype=torch.float64,
                 device="cuda", max_batch=1, kernel="auto", n_extra=0, rolling_window=1, forward_rolling=True,
                 activations=None):
        if not torch.cuda.is_available():
            raise RuntimeError("pyneuralempc_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("CallbackEngine runs on a HIP device only")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if dtype not in _TORCH_DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.dtype = dtype
        # rolling-window models (model/tensorflow.py:132, model/jax.py:93): the network of step t reads the last
        # `rolling_window` states and controls; nin is the tile width = number of decision inputs one step reads
        self.rolling_window = int(rolling_window)
        if self.rolling_window < 1:
            raise ValueError("Your rolling windows need to be an integer gretter than 1.")
        self.forward_rolling = bool(forward_rolling)
        self._history = None
        self.H, self.nx, self.nu = int(H), int(nx), int(nu)
        self.nin = self.rolling_window * (self.nx + self.nu)
        self.n_extra = int(n_extra)   # tvp_dim + p_dim of the reference's Model: network inputs that are not variables
        self._extra = None
        self.integrator = integrator if isinstance(integrator, int) else _lib.INTEGRATOR_IDS[integrator]
        self.DT = float(DT)
        self.kernel = kernel
        self._weights = [np.ascontiguousarray(w, dtype=np.float64) for w in weights]
        self._biases = [np.ascontiguousarray(b, dtype=np.float64).reshape(-1) for b in biases]
        if len(self._weights) != len(self._biases) or not self._weights:
            raise ValueError("weights and biases must be non-empty lists of equal length")
        if len(self._weights) > _lib.MAX_LAYERS:
            raise ValueError(f"at most {_lib.MAX_LAYERS} dense layers are supported")
        prev = self.nin + self.n_extra
        for w, b in zip(self._weights, self._biases):
            if w.ndim != 2 or w.shape[0] != prev or b.shape != (w.shape[1],):
                raise ValueError("layer shapes do not chain: expected kernel (in,out) and bias (out,)")
            prev = w.shape[1]
        if prev != self.nx:
            raise ValueError("Your model do not provide a suitable output dim ! It must get the same dim as the "
                             "state dim.")
        self.activations = resolve_activations(activations, len(self._weights))
        self._objective = None
        self._box = None
        self._handle = None
        self.max_batch = 0
        self._buffers = {}
        self._create(int(max_batch))

    # ------------------------------------------------------------------ handle management
    def _create(self, max_batch):
        self._destroy()
        cfg = _lib.NempcConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.device = self.device.index
        cfg.dtype = _TORCH_DTYPES[self.dtype]
        cfg.integrator = self.integrator
        cfg.H, cfg.nx, cfg.nu = self.H, self.nx, self.nu
        cfg.n_layers = len(self._weights)
        for i, w in enumerate(self._weights):
            cfg.widths[i] = w.shape[1]
            name, par = _lib.split_activation(self.activations[i])
            cfg.activations[i], cfg.act_param[i] = _lib.ACTIVATION_IDS[name], par
        cfg.max_batch = max_batch
        cfg.kernel = _lib.KERNEL_NAMES[self.kernel] if isinstance(self.kernel, str) else int(self.kernel)
        cfg.n_extra = self.n_extra
        cfg.rolling_window = self.rolling_window
        cfg.rolling_reverse = 0 if self.forward_rolling else 1
        cfg.DT = self.DT
        h = ctypes.c_void_p()
        _lib.check(self.lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._handle = h
        self.max_batch = max_batch
        nl = len(self._weights)
        dp = ctypes.POINTER(ctypes.c_double)
        Wp = (dp * nl)(*[w.ctypes.data_as(dp) for w in self._weights])
        bp = (dp * nl)(*[b.ctypes.data_as(dp) for b in self._biases])
        _lib.check(self.lib.nempc_set_weights(self._handle, Wp, bp))
        if self._objective is not None:
            self.set_objective(**self._objective)
        if self._box is not None:
            self.set_box_rows(*self._box)
        if self._extra is not None:
            _lib.check(self.lib.nempc_bind_extra(self._handle, ctypes.c_void_p(self._extra.data_ptr()),
                                                 int(self._extra.shape[0])))
        if self._history is not None:
            self.bind_history(*self._history)
        self._refresh_dims()
        self._buffers = {}

    def _destroy(self):
        if getattr(self, "_handle", None):
            self.lib.nempc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def __getstate__(self):
        raise TypeError("CallbackEngine holds device handles and is not picklable; pickle the plugin objects")

    def _refresh_dims(self):
        n, m, nj, nh = (ctypes.c_int32() for _ in range(4))
        _lib.check(self.lib.nempc_dims(self._handle, ctypes.byref(n), ctypes.byref(m), ctypes.byref(nj),
                                       ctypes.byref(nh)))
        self.n, self.m, self.nnz_jac, self.nnz_hess = n.value, m.value, nj.value, nh.value

    def reserve(self, B):
        """Grow the handle's workspaces to hold B problems (nempc_reserve: the handle, its bindings and its
        communicator are kept)."""
        if B > self.max_batch:
            _lib.check(self.lib.nempc_reserve(self._handle, int(B)))
            self.max_batch = int(B)
            self._buffers = {}

    @property
    def kernel_variant(self):
        v = self.lib.nempc_kernel_variant(self._handle)
        return {_lib.KERNEL_VALU: "valu", _lib.KERNEL_MFMA: "mfma", _lib.KERNEL_MFMA_TILE: "mfma_tile",
                _lib.KERNEL_LAYERED: "layered"}[v]

    @property
    def last_row_kernel(self):
        """Name of the row kernel the most recent evaluation launched."""
        return {0: None, 1: "rows_valu_kernel", 2: "rows_coop_kernel", 3: "rows_mfma_kernel", 4: "rows_coopfx_kernel",
                5: "rows_coop_kernel+dense", 6: "rows_coopfx_kernel+sparse", 7: "rows_coop_kernel+sparse", 8: "layered_gemm_kernel"}[
            self.lib.nempc_last_row_kernel(self._handle)]

    @property
    def last_hess_kernel(self):
        """Name of the network kernel the most recent Lagrangian-Hessian evaluation launched ("rk4:" prefix: inside the RK4
        pipeline)."""
        v = self.lib.nempc_last_hess_kernel(self._handle)
        name = {0: None, 1: "rowhess_valu_kernel", 2: "rowhess_coop_kernel", 3: "rowhess_mfma_kernel", 4: "rowhess_coopfx_kernel",
                5: "layered_gemm_kernel"}[v % 10]
        return ("rk4:" + name) if v >= 10 and name else name

    # ------------------------------------------------------------------ parameters
    def set_objective(self, Q=None, R=None, xref=None, uref=None, cx=None, cu=None, QT=None):
        """Quadratic + linear stage cost; QT (nx,nx) is the state weight of the last step (terminal cost), None = Q."""
        H, nx, nu = self.H, self.nx, self.nu
        keep, ptrs = [], []
        for v, shape in ((Q, (nx, nx)), (R, (nu, nu)), (xref, (H, nx)), (uref, (H, nu)), (cx, (H, nx)),
                         (cu, (H, nu))):
            if v is None:
                ptrs.append(None)
            else:
                a, p = _as_c_double(np.broadcast_to(np.asarray(v, dtype=np.float64), shape))
                keep.append(a)
                ptrs.append(p)
        _lib.check(self.lib.nempc_set_objective(self._handle, *ptrs))
        if QT is None:
            _lib.check(self.lib.nempc_set_terminal_weight(self._handle, None))
        else:
            a, pt = _as_c_double(np.asarray(QT, dtype=np.float64).reshape(nx, nx))
            _lib.check(self.lib.nempc_set_terminal_weight(self._handle, pt))
        self._objective = dict(Q=Q, R=R, xref=xref, uref=uref, cx=cx, cu=cu, QT=QT)
        self._refresh_dims()

    def bind_extra(self, E):
        """Bind the extra network inputs (B,H,n_extra) = [tvp_t ; p] per step for the following evaluations
        (reference: KerasTFModel._gather_input, model/tensorflow.py:39-47).  The tensor is kept alive here."""
        if self.n_extra == 0:
            raise ValueError("this engine was created with n_extra = 0")
        if E.device != self.device or E.dtype != self.dtype or E.dim() != 3 or tuple(E.shape[1:]) != (self.H, self.n_extra):
            raise ValueError(f"extra inputs must be a {self.dtype} tensor (B,{self.H},{self.n_extra}) on {self.device}")
        self._extra = E.contiguous()
        _lib.check(self.lib.nempc_bind_extra(self._handle, ctypes.c_void_p(self._extra.data_ptr()),
                                             int(self._extra.shape[0])))

    def bind_history(self, hist_x, hist_u):
        """Bind the history of a rolling-window model: hist_x (B, w-1, nx) states before x0 (oldest first), hist_u
        (B, w-1, nu) controls before u_0 -- the batched form of set_prev_data (model/tensorflow.py:178-189)."""
        back = self.rolling_window - 1
        if back == 0:
            raise ValueError("this engine was created with rolling_window = 1")
        for t, d, name in ((hist_x, self.nx, "hist_x"), (hist_u, self.nu, "hist_u")):
            if (not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != self.dtype or t.dim() != 3
                    or tuple(t.shape[1:]) != (back, d)):
                raise ValueError(f"{name} must be a {self.dtype} tensor (B,{back},{d}) on {self.device}")
        if hist_x.shape[0] != hist_u.shape[0]:
            raise ValueError("hist_x and hist_u must cover the same batch")
        self._history = (hist_x.contiguous(), hist_u.contiguous())
        _lib.check(self.lib.nempc_bind_history(self._handle, ctypes.c_void_p(self._history[0].data_ptr()),
                                               ctypes.c_void_p(self._history[1].data_ptr()),
                                               int(self._history[0].shape[0])))

    def _check_extra(self, B):
        if self.n_extra and (self._extra is None or self._extra.shape[0] < B):
            raise ValueError("n_extra > 0: call bind_extra with a (B,H,n_extra) tensor covering the batch first")
        if self.rolling_window > 1 and (self._history is None or self._history[0].shape[0] < B):
            raise ValueError("You must give history window with set_prev_data before calling any inferance function.")

    def set_box_rows(self, lo, hi):
        if lo is None:
            _lib.check(self.lib.nempc_set_box_rows(self._handle, 0, None, None))
            self._box = None
        else:
            a, pa = _as_c_double(np.broadcast_to(np.asarray(lo, dtype=np.float64), (self.nx,)))
            b, pb = _as_c_double(np.broadcast_to(np.asarray(hi, dtype=np.float64), (self.nx,)))
            _lib.check(self.lib.nempc_set_box_rows(self._handle, 1, pa, pb))
            self._box = (lo, hi)
        self._refresh_dims()
        self._buffers = {}

    # ------------------------------------------------------------------ structure / bounds (host)
    def constraint_bounds(self):
        cl, cu = np.empty(self.m), np.empty(self.m)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.check(self.lib.nempc_constraint_bounds(self._handle, cl.ctypes.data_as(dp), cu.ctypes.data_as(dp)))
        return cl, cu

    def jac_structure(self):
        r, c = np.empty(self.nnz_jac, dtype=np.int32), np.empty(self.nnz_jac, dtype=np.int32)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.check(self.lib.nempc_jac_structure(self._handle, r.ctypes.data_as(ip), c.ctypes.data_as(ip)))
        return r, c

    def hess_structure(self):
        r, c = np.empty(self.nnz_hess, dtype=np.int32), np.empty(self.nnz_hess, dtype=np.int32)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.check(self.lib.nempc_hess_structure(self._handle, r.ctypes.data_as(ip), c.ctypes.data_as(ip)))
        return r, c

    # ------------------------------------------------------------------ evaluation
    def _check_in(self, t, shape, name):
        if not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != self.dtype:
            raise ValueError(f"{name} must be a {self.dtype} tensor on {self.device}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")

    def _out(self, name, shape):
        key = (name, tuple(shape))
        buf = self._buffers.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=self.dtype, device=self.device)
            self._buffers[key] = buf
        return buf

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def eval(self, Z, X0, want=("f", "grad", "g", "jac_dense"), out=None):
        """One batched callback evaluation.  Z (B,n), X0 (B,nx) device tensors.  Returns a dict with
        the requested outputs (device tensors, reused between calls unless `out` supplies them):
        f (B,), grad (B,n), g (B,m), jac_dense (B,m,n), jac_tiles (B,H,nx,w*(nx+nu)), jac_sparse (B,nnz_jac).
        Asynchronous on the current torch stream."""
        B = int(Z.shape[0])
        self._check_in(Z, (B, self.n), "Z")
        self._check_in(X0, (B, self.nx), "X0")
        self._check_extra(B)
        self.reserve(B)
        shapes = {"f": (B,), "grad": (B, self.n), "g": (B, self.m), "jac_dense": (B, self.m, self.n),
                  "jac_tiles": (B, self.H, self.nx, self.nin), "jac_sparse": (B, self.nnz_jac)}
        res, ptr = {}, {}
        for k in shapes:
            if k in want:
                t = out[k] if (out is not None and k in out) else self._out(k, shapes[k])
                self._check_in(t, shapes[k], k)
                res[k] = t
                ptr[k] = ctypes.c_void_p(t.data_ptr())
            else:
                ptr[k] = None
        unknown = set(want) - set(shapes)
        if unknown:
            raise ValueError(f"unknown outputs requested: {sorted(unknown)}")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_eval(self._handle, B, ctypes.c_void_p(Z.data_ptr()),
                                           ctypes.c_void_p(X0.data_ptr()), ptr["f"], ptr["grad"], ptr["g"],
                                           ptr["jac_dense"], ptr["jac_tiles"], ptr["jac_sparse"], self._stream()))
        return res

    def bind(self, Z, X0, want=("f", "grad", "g", "jac_dense"), out=None):
        """Pre-validated evaluation for hot loops: returns (launch, outputs) where launch() re-evaluates the
        callbacks at the CURRENT contents of Z / X0 into the fixed `outputs` tensors (caller-supplied through `out`,
        else the engine's reusable buffers) with one ctypes call (no per-call allocation or checking).  Valid while
        Z, X0 and the outputs stay alive and the workspaces are not re-sized (reserve / set_box_rows)."""
        B = int(Z.shape[0])
        res = self.eval(Z, X0, want, out=out)             # validates, allocates outputs, warms the kernels
        ptr = {k: (ctypes.c_void_p(res[k].data_ptr()) if k in res else None)
               for k in ("f", "grad", "g", "jac_dense", "jac_tiles", "jac_sparse")}
        handle, fn = self._handle, self.lib.nempc_eval
        zp, xp = ctypes.c_void_p(Z.data_ptr()), ctypes.c_void_p(X0.data_ptr())
        stream = self._stream()
        args = (handle, B, zp, xp, ptr["f"], ptr["grad"], ptr["g"], ptr["jac_dense"], ptr["jac_tiles"],
                ptr["jac_sparse"], stream)

        def launch():
            rc = fn(*args)
            if rc:
                _lib.check(rc)
        return launch, res

    def hess(self, Z, X0, lam, sigma, want=("hvals",)):
        """Lagrangian Hessian: hvals (B,nnz_hess) in hess_structure() order, optional hdense (B,n,n)
        and hblocks (B,H,w*(nx+nu),w*(nx+nu))."""
        B = int(Z.shape[0])
        self._check_in(Z, (B, self.n), "Z")
        self._check_in(X0, (B, self.nx), "X0")
        self._check_in(lam, (B, self.m), "lam")
        self._check_in(sigma, (B,), "sigma")
        self._check_extra(B)
        self.reserve(B)
        shapes = {"hvals": (B, self.nnz_hess), "hdense": (B, self.n, self.n),
                  "hblocks": (B, self.H, self.nin, self.nin)}
        res, ptr = {}, {}
        for k in shapes:
            if k in want:
                res[k] = self._out(k, shapes[k])
                ptr[k] = ctypes.c_void_p(res[k].data_ptr())
            else:
                ptr[k] = None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_hess(self._handle, B, ctypes.c_void_p(Z.data_ptr()),
                                           ctypes.c_void_p(X0.data_ptr()), ctypes.c_void_p(lam.data_ptr()),
                                           ctypes.c_void_p(sigma.data_ptr()), ptr["hvals"], ptr["hdense"],
                                           ptr["hblocks"], self._stream()))
        return res

    def hess_gn(self, Z, X0, w=None, sigma=None, want=("hvals",)):
        """Gauss-Newton Hessian (nempc_hess_gn): sigma * d2f + sum_t T_t^T diag(w_t) T_t in hess_structure() order, from
        the first-order tiles only.  w (B, H*nx) weights per defect row (None = ones), sigma (B,) (None = ones)."""
        B = int(Z.shape[0])
        self._check_in(Z, (B, self.n), "Z")
        self._check_in(X0, (B, self.nx), "X0")
        if w is not None:
            self._check_in(w, (B, self.H * self.nx), "w")
        if sigma is None:
            sigma = torch.ones(B, dtype=self.dtype, device=self.device)
        self._check_in(sigma, (B,), "sigma")
        self._check_extra(B)
        self.reserve(B)
        shapes = {"hvals": (B, self.nnz_hess), "hdense": (B, self.n, self.n),
                  "hblocks": (B, self.H, self.nin, self.nin)}
        res, ptr = {}, {}
        for k in shapes:
            if k in want:
                res[k] = self._out("gn_" + k, shapes[k])
                ptr[k] = ctypes.c_void_p(res[k].data_ptr())
            else:
                ptr[k] = None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_hess_gn(self._handle, B, ctypes.c_void_p(Z.data_ptr()),
                                              ctypes.c_void_p(X0.data_ptr()),
                                              None if w is None else ctypes.c_void_p(w.data_ptr()),
                                              ctypes.c_void_p(sigma.data_ptr()), ptr["hvals"], ptr["hdense"],
                                              ptr["hblocks"], self._stream()))
        return res

    def bind_hess(self, Z, X0, lam, sigma, want=("hvals",), gauss_newton=False):
        """Pre-validated Hessian callback for hot loops (the counterpart of bind): returns (launch, outputs); launch()
        re-evaluates the callback at the CURRENT contents of Z / X0 / lam / sigma into the fixed outputs with one ctypes
        call.  gauss_newton=True binds nempc_hess_gn, `lam` then being the (B, H*nx) row weights or None."""
        B = int(Z.shape[0])
        res = self.hess_gn(Z, X0, lam, sigma, want) if gauss_newton else self.hess(Z, X0, lam, sigma, want)
        if gauss_newton and sigma is None:
            raise ValueError("bind_hess: pass sigma explicitly (the bound call reads the tensor every time)")
        ptr = {k: (ctypes.c_void_p(res[k].data_ptr()) if k in res else None) for k in ("hvals", "hdense", "hblocks")}
        fn = self.lib.nempc_hess_gn if gauss_newton else self.lib.nempc_hess
        args = (self._handle, B, ctypes.c_void_p(Z.data_ptr()), ctypes.c_void_p(X0.data_ptr()),
                None if lam is None else ctypes.c_void_p(lam.data_ptr()), ctypes.c_void_p(sigma.data_ptr()),
                ptr["hvals"], ptr["hdense"], ptr["hblocks"], self._stream())

        def launch():
            rc = fn(*args)
            if rc:
                _lib.check(rc)
        return launch, res

    def solve(self, X0, Z_init=None, lb=None, ub=None, max_iter=200, max_linesearch=6, check_every=4,
              tol_constraint=None, tol_step=None, mu_init=1e-1, mu_min=None, mu_factor=0.2, reg=None, lq_kernel="auto",
              compact=True, return_iterations=False, barrier="primal-dual", linesearch="auto", lq_attempts=0):
        """Batched on-device solve (Gauss-Newton SQP, see csrc/solver.hip).  X0 (B,nx) device tensor; Z_init (B,n)
        or None for the reference's cold start [x0 tiled H ; zeros] (optimizer/ipopt.py:149); lb/ub (n) host
        vectors as DomainConstraint produces them; tolerances default by dtype (fp64 1e-8, fp32 1e-4); lq_kernel picks the LQ solve ("auto" | "thread" per problem | "wave"
        per problem | "scan": parallel in time, 2/1 stages in fp64); compact=True gathers the unconverged problems to the front as the batch converges (same results,
        shorter launches); linesearch "loop" backtracks inside an iteration (every problem waits for the slowest search),
        "deferred" tries one step length per iteration and lets a rejected problem retry at half the length in the next one
        (about half the time per iteration at the 2/1 shape, more iterations for hard problems), "auto" defers for small
        stages and loops for matrix-core-bound ones; lq_attempts bounds the Riccati sweeps a problem may try per iteration
        (0: the library's default of 3).  Returns (Z (B,n), status (B,) int32 [0 ok / 1 fail], iters) [+ per-problem convergence
        iteration (B,) int32 with return_iterations=True]."""
        B = int(X0.shape[0])
        self._check_in(X0, (B, self.nx), "X0")
        self._check_extra(B)
        self.reserve(B)
        # tolerances an iterate of the handle's precision can reach (fp32 stalls near 1e-5 relative)
        f64 = self.dtype == torch.float64
        tol_constraint = (1e-8 if f64 else 1e-4) if tol_constraint is None else tol_constraint
        tol_step = (1e-8 if f64 else 1e-4) if tol_step is None else tol_step
        mu_min = (1e-9 if f64 else 1e-5) if mu_min is None else mu_min
        reg = (1e-9 if f64 else 1e-6) if reg is None else reg
        if Z_init is None:
            Z = torch.cat([X0.repeat(1, self.H), torch.zeros(B, self.H * self.nu, dtype=self.dtype, device=self.device)],
                          dim=1).contiguous()
        else:
            self._check_in(Z_init, (B, self.n), "Z_init")
            Z = Z_init.clone()
        keep, ptrs = [], []
        for v in (lb, ub):
            if v is None:
                ptrs.append(None)
            else:
                a, p = _as_c_double(np.broadcast_to(np.asarray(v, dtype=np.float64), (self.n,)))
                keep.append(a)
                ptrs.append(p)
        its_dev = torch.zeros(B, dtype=torch.int32, device=self.device) if return_iterations else None
        opts = _lib.NempcSolverOpts(max_iter=max_iter, max_linesearch=max_linesearch, check_every=check_every,
                                    lq_kernel={"auto": 0, "thread": 1, "wave": 2, "scan": 3}[lq_kernel],
                                    tol_constraint=tol_constraint, tol_step=tol_step, mu_init=mu_init, mu_min=mu_min,
                                    mu_factor=mu_factor, reg=reg, compact=1 if compact else 0,
                                    barrier={"primal-dual": 0, "primal": 1}[barrier],
                                    iters_out=None if its_dev is None else its_dev.data_ptr(),
                                    linesearch={"auto": 0, "loop": 1, "deferred": 2}[linesearch], lq_attempts=int(lq_attempts))
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        iters = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_solve(self._handle, B, ctypes.c_void_p(X0.data_ptr()), ctypes.c_void_p(Z.data_ptr()),
                                            ptrs[0], ptrs[1], ctypes.byref(opts), ctypes.c_void_p(status.data_ptr()),
                                            ctypes.byref(iters), self._stream()))
        if return_iterations:
            return Z, status, iters.value, its_dev
        return Z, status, iters.value

    def sync(self):
        _lib.check(self.lib.nempc_sync(self._handle, self._stream()))

    # ------------------------------------------------------------------ multi-GPU: the u0 all-gather on RCCL
    def comm_init(self, nranks, rank, unique_id):
        """Collective over the ranks: build this handle's RCCL communicator from the 128-byte id rank 0 obtained
        with `CallbackEngine.comm_unique_id()` (parallel.init_u0_comm moves it over torch.distributed)."""
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.COMM_ID_BYTES} bytes")
        buf = (ctypes.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_comm_init(self._handle, int(nranks), int(rank), ctypes.cast(buf, ctypes.c_void_p)))
        self._comm = (int(nranks), int(rank))

    @staticmethod
    def comm_unique_id():
        buf = (ctypes.c_char * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.load().nempc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    @property
    def comm(self):
        """(nranks, rank) of the handle's communicator, or None."""
        return getattr(self, "_comm", None)

    def allgather_u0(self, Z=None, u0=None, rows_per_rank=None, out=None):
        """nempc_allgather_u0: first controls of this rank's B problems -- read from Z (B,n) or given as u0 (B,nu) --
        gathered over the communicator into (nranks*rows_per_rank, nu), rank-major; asynchronous on the current
        stream.  rows_per_rank defaults to B (equal shards); ragged shards pass the largest shard size."""
        if self.comm is None:
            raise RuntimeError("allgather_u0: call comm_init first")
        if (Z is None) == (u0 is None):
            raise ValueError("pass exactly one of Z and u0")
        src = Z if Z is not None else u0
        B = int(src.shape[0])
        self._check_in(src, (B, self.n if Z is not None else self.nu), "Z" if Z is not None else "u0")
        rows = B if rows_per_rank is None else int(rows_per_rank)
        shape = (self.comm[0] * rows, self.nu)
        if out is None:
            out = self._out("u0_gathered", shape)
        self._check_in(out, shape, "out")
        zp = ctypes.c_void_p(Z.data_ptr()) if Z is not None else None
        up = ctypes.c_void_p(u0.data_ptr()) if u0 is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nempc_allgather_u0(self._handle, B, rows, zp, up, ctypes.c_void_p(out.data_ptr()),
                                                   self._stream()))
        return out

    def bind_allgather_u0(self, Z, stream=None, rows_per_rank=None):
        """Pre-validated all-gather for hot loops (the counterpart of `bind`): returns (launch, gathered) where launch()
        re-issues nempc_allgather_u0 on `stream` (a torch.cuda.Stream; default: the current one) at the CURRENT contents
        of Z with one ctypes call."""
        out = self.allgather_u0(Z=Z, rows_per_rank=rows_per_rank)          # validates, allocates, warms
        B = int(Z.shape[0])
        rows = B if rows_per_rank is None else int(rows_per_rank)
        st = ctypes.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)
        args = (self._handle, B, rows, ctypes.c_void_p(Z.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), st)
        fn = self.lib.nempc_allgather_u0

        def launch():
            rc = fn(*args)
            if rc:
                _lib.check(rc)
        return launch, out

    # ------------------------------------------------------------------ host convenience (B=1 drop-in path)
    def to_device(self, a):
        # (a read-only view -- np.broadcast_to -- is copied: torch refuses to alias memory it may not write)
        return torch.as_tensor(np.require(a, dtype=np.float64, requirements=["C", "W"])).to(self.device, self.dtype)

    def eval_numpy(self, Z, X0, want=("f", "grad", "g", "jac_dense")):
        Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
        X0 = np.atleast_2d(np.asarray(X0, dtype=np.float64))
        res = self.eval(self.to_device(Z), self.to_device(X0), want)
        return {k: v.to("cpu", torch.float64).numpy() for k, v in res.items()}
