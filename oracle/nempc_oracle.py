"""CPU oracle for the NMPC callback path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  The product path (``pyneuralempc_amd``) never does; it goes through
the C-ABI HIP library and fails loudly when that library is missing.

What it is: an fp64 NumPy restatement of the per-iterate evaluation the reference hands to its
NLP solver -- ``f, grad f, g, jac g`` and the Lagrangian Hessian -- for a feed-forward MLP
dynamics model (tanh by default; the activation family of `ACTIVATIONS`) under the Discret / Unity / RK4 transcriptions.  File:line citations are into
``/root/reference/pyNeuralEMPC``.

Parity pinning (see DESIGN.md "Oracle"):
  * integrator / glue / bounds / warm-start arithmetic: pinned against the *imported reference
    modules themselves* (integrator/{discret,unity,rk4}.py, constraints.py, optimizer/slsqp.py,
    controller.py) through the golden vectors in ``tests/golden`` produced by
    ``tests/golden/make_golden.py``.
  * the derivative of the network itself is delegated by the reference to TensorFlow / JAX
    autodiff (model/tensorflow.py:53-75, model/jax.py:52-62), which are un-vendored, un-pinned
    (setup.py:20) and not installed here.  That piece is restated analytically (chain rule of a
    tanh MLP) and pinned by an independent AD (``torch.func`` fp64) and by finite differences
    in ``tests/test_oracle.py``.  The reference ships no test or fixture for it.

Layouts (all row-major, identical to the reference):
  z        (n,)      n = H*(nx+nu);  z[:H*nx] = states (H,nx),  z[H*nx:] = controls (H,nu)
                     optimizer/ipopt.py:20-28
  x_prev   (H,nx)    [x0 ; states[:-1]]                               integrator/discret.py:22
  g        (m,)      integrator defects (H*nx) then each extra constraint in list order
                     optimizer/ipopt.py:44-52
  jac      (m,n)     dense                                           optimizer/ipopt.py:88-96
"""
from __future__ import annotations

import numpy as np

DISCRET, UNITY, RK4 = 0, 1, 2
INTEGRATOR_NAMES = {DISCRET: "discret", UNITY: "unity", RK4: "rk4"}


# --------------------------------------------------------------------------------------
# Activations.  The reference wraps ANY feed-forward Keras model (model/tensorflow.py:8-29,49-51) or JAX callable
# (model/jax.py:32-49) and lets TensorFlow / JAX differentiate it; the family below is what Keras Dense layers are
# commonly built with.  Every derivative is written in terms of the layer's OUTPUT a = s(z), the value the forward pass
# keeps (the device kernels keep nothing else):
#     s'(z) = d1(a),      s''(z) = r2(a) * d1(a)
#   linear    a = z                     d1 = 1                       r2 = 0
#   tanh      a = tanh z                d1 = 1 - a^2                 r2 = -2a
#   relu      a = max(z, 0)             d1 = [a > 0]                 r2 = 0           (TF's relu gradient at 0 is 0)
#   sigmoid   a = 1/(1+e^-z)            d1 = a(1-a)                  r2 = 1 - 2a
#   softplus  a = log(1+e^z)            d1 = 1 - e^-a (= sigmoid z)  r2 = e^-a (= 1 - sigmoid z)
#   elu       a = z (z>0), al(e^z-1)    d1 = 1 (a>0), a+al (a<=0)    r2 = 0 (a>0), 1 (a<=0)   (al = alpha > 0, Keras default 1)
#   leaky_relu a = z (z>0), al z        d1 = 1 (a>0), al (a<=0)      r2 = 0           (al >= 0; tf.nn.leaky_relu's gradient at 0 is al)
#   selu      a = la z (z>0), la al(e^z-1)  d1 = la (a>0), a + la al  r2 = 0 (a>0), 1  (la, al: Keras' fixed constants)
# The monotone activations above are the ones whose derivatives follow from the output alone; swish / gelu (not monotone)
# would need the pre-activation and are not part of the family.  A parameterised activation is written "name:value"
# ("elu:0.5", "leaky_relu:0.1"); the bare name takes the default (elu 1, leaky_relu 0.2 = keras.activations.leaky_relu).
# --------------------------------------------------------------------------------------
ACTIVATIONS = ("linear", "tanh", "relu", "sigmoid", "softplus", "elu", "leaky_relu", "selu", "swish", "gelu", "softsign", "mish",
               "exponential", "relu6")
ZBASED = ("swish", "gelu", "softsign", "mish", "exponential", "relu6")      # s', s'' written from the pre-activation z (act_s1 / act_s2)
ACT_IDS = {name: i for i, name in enumerate(ACTIVATIONS)}     # the codes of include/nempc.h (NEMPC_ACT_*)
ACT_DEFAULT_PARAM = {"elu": 1.0, "leaky_relu": 0.2}
SELU_LAMBDA, SELU_ALPHA = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717


def act_split(spec):
    """"name" or "name:value" -> (name, parameter); the parameter is alpha of elu / leaky_relu, 0.0 for the others"""
    name, _, val = str(spec).partition(":")
    if name not in ACTIVATIONS:
        raise ValueError(f"unknown activation {spec!r}")
    if val and name not in ACT_DEFAULT_PARAM:
        raise ValueError(f"activation {name!r} takes no parameter ({spec!r})")
    par = float(val) if val else ACT_DEFAULT_PARAM.get(name, 0.0)
    if name == "elu" and not par > 0.0:
        raise ValueError("elu needs alpha > 0 (its derivative is written from the output)")
    if name == "leaky_relu" and not par >= 0.0:
        raise ValueError("leaky_relu needs alpha >= 0")
    return name, par


def act_f(name, z):
    name, par = act_split(name)
    if name == "linear":
        return z
    if name == "tanh":
        return np.tanh(z)
    if name == "relu":
        return np.where(z < 0.0, 0.0, z)              # (a NaN stays a NaN)
    if name == "sigmoid":
        with np.errstate(over="ignore"):
            return 1.0 / (1.0 + np.exp(-z))
    if name == "softplus":
        return np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z)))
    if name == "elu":
        return np.where(z > 0.0, z, par * np.expm1(np.minimum(z, 0.0)))
    if name == "leaky_relu":
        return np.where(z > 0.0, z, par * z)
    if name == "selu":
        return SELU_LAMBDA * np.where(z > 0.0, z, SELU_ALPHA * np.expm1(np.minimum(z, 0.0)))
    if name == "swish":                                # Keras swish / silu: z sigmoid(z)
        with np.errstate(over="ignore"):
            return z / (1.0 + np.exp(-z))
    if name == "gelu":                                 # Keras gelu, approximate=False: z Phi(z)
        from scipy.special import erf
        return 0.5 * z * (1.0 + erf(z / np.sqrt(2.0)))
    if name == "softsign":
        return z / (1.0 + np.abs(z))
    if name == "mish":                                 # z tanh(softplus(z))
        return z * np.tanh(np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z))))
    if name == "exponential":
        with np.errstate(over="ignore"):
            return np.exp(z)
    if name == "relu6":                                # Keras ReLU(max_value=6)
        return np.where(z < 0.0, 0.0, np.where(z > 6.0, 6.0, z))
    raise ValueError(f"unknown activation {name!r}")


def act_s1(name, z, a):
    """s'(z) of a layer with pre-activation z and output a: from the output for the monotone activations (act_d1: what
    the register-resident kernels do), from z for swish / gelu"""
    nm = act_split(name)[0]
    if nm == "swish":
        with np.errstate(over="ignore"):
            sg = 1.0 / (1.0 + np.exp(-z))
        return sg * (1.0 + z * (1.0 - sg))
    if nm == "gelu":
        from scipy.special import erf
        return 0.5 * (1.0 + erf(z / np.sqrt(2.0))) + z * np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)
    if nm == "softsign":
        return 1.0 / (1.0 + np.abs(z)) ** 2
    if nm == "mish":
        t, g = _mish_parts(z)
        return t + z * (1.0 - t * t) * g
    if nm == "exponential":
        return a
    if nm == "relu6":
        return np.where(np.isnan(z), z, ((z > 0.0) & (z < 6.0)).astype(np.float64))
    return act_d1(name, a)


def act_s2(name, z, a):
    """s''(z), same convention"""
    nm = act_split(name)[0]
    if nm == "swish":
        with np.errstate(over="ignore"):
            sg = 1.0 / (1.0 + np.exp(-z))
        return sg * (1.0 - sg) * (2.0 + z * (1.0 - 2.0 * sg))
    if nm == "gelu":
        return np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi) * (2.0 - z * z)
    if nm == "softsign":
        return -2.0 * np.sign(z) / (1.0 + np.abs(z)) ** 3
    if nm == "mish":
        t, g = _mish_parts(z)
        u = (1.0 - t * t) * g
        return 2.0 * u + z * u * ((1.0 - g) - 2.0 * t * g)
    if nm == "exponential":
        return a
    if nm == "relu6":
        return np.where(np.isnan(z), z, 0.0)
    return act_r2(name, a) * act_d1(name, a)


def _mish_parts(z):
    """t = tanh(softplus z), g = sigmoid z"""
    with np.errstate(over="ignore"):
        return np.tanh(np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z)))), 1.0 / (1.0 + np.exp(-z))


def act_d1(name, a):
    name, par = act_split(name)
    if name == "linear":
        return np.ones_like(a)
    if name == "tanh":
        return 1.0 - a * a
    if name == "relu":
        return (a > 0.0).astype(np.float64)
    if name == "sigmoid":
        return a * (1.0 - a)
    if name == "softplus":
        return -np.expm1(-a)
    if name == "elu":
        return np.where(a > 0.0, 1.0, a + par)
    if name == "leaky_relu":
        return np.where(a > 0.0, 1.0, np.where(np.isnan(a), a, par))
    if name == "selu":
        return np.where(a > 0.0, SELU_LAMBDA, a + SELU_LAMBDA * SELU_ALPHA)
    raise ValueError(f"unknown activation {name!r}")


def act_r2(name, a):
    name, par = act_split(name)
    if name in ("linear", "relu", "leaky_relu"):
        return np.zeros_like(a)
    if name == "tanh":
        return -2.0 * a
    if name == "sigmoid":
        return 1.0 - 2.0 * a
    if name == "softplus":
        return np.exp(-a)
    if name in ("elu", "selu"):
        return np.where(a > 0.0, 0.0, 1.0)
    raise ValueError(f"unknown activation {name!r}")


# --------------------------------------------------------------------------------------
# Network: a_0 = xi ; a_l = s_l(a_{l-1} W_l + b_l)  (Keras kernel (in,out)); default s = tanh on the hidden layers and a
# linear output layer (the reference's own nn_model.h5).  Restates what model/tensorflow.py:49-51 evaluates with
# model.predict on [x | u] rows
# --------------------------------------------------------------------------------------
class MLP:
    def __init__(self, weights, biases, activations=None):
        self.W = [np.asarray(w, dtype=np.float64) for w in weights]
        self.b = [np.asarray(b, dtype=np.float64) for b in biases]
        assert len(self.W) == len(self.b) and len(self.W) >= 1
        for w, b in zip(self.W, self.b):
            assert w.ndim == 2 and b.shape == (w.shape[1],)
        for w0, w1 in zip(self.W[:-1], self.W[1:]):
            assert w0.shape[1] == w1.shape[0]
        self.n_in = self.W[0].shape[0]
        self.n_out = self.W[-1].shape[1]
        if activations is None:
            activations = ["tanh"] * (len(self.W) - 1) + ["linear"]
        elif isinstance(activations, str):             # one name: every hidden layer, linear output
            activations = [activations] * (len(self.W) - 1) + ["linear"]
        self.act = [str(a) for a in activations]
        assert len(self.act) == len(self.W)
        for a in self.act:
            act_split(a)

    @staticmethod
    def random(n_in, hidden, n_out, seed=0, activations=None):
        """SURVEY.md 8(d) synthetic weights: W ~ N(0, 1/fan_in), b ~ N(0, 0.1^2)."""
        rng = np.random.default_rng(seed)
        dims = [n_in] + list(hidden) + [n_out]
        W, b = [], []
        for i, o in zip(dims[:-1], dims[1:]):
            W.append(rng.normal(0.0, 1.0 / np.sqrt(i), size=(i, o)))
            b.append(rng.normal(0.0, 0.1, size=(o,)))
        return MLP(W, b, activations)

    def _acts(self, xi, with_z=False):
        """Outputs of every layer, acts[0] = the input, acts[l+1] = output of layer l (post-activation); with_z: also the
        pre-activations zs[l]"""
        acts, zs = [np.asarray(xi, dtype=np.float64)], []
        for w, b, name in zip(self.W, self.b, self.act):
            zs.append(acts[-1] @ w + b)
            acts.append(act_f(name, zs[-1]))
        return (acts, zs) if with_z else acts

    def forward(self, xi):
        """(R, n_in) -> (R, n_out)"""
        return self._acts(xi)[-1]

    def forward_jac(self, xi):
        """(R, n_in) -> f (R, n_out), J (R, n_out, n_in).

        Reverse sweep: J = D_L W_L^T D_{L-1} W_{L-1}^T ... D_1 W_1^T with D_l = diag(s_l'(z_l)) (the identity for a
        linear layer), the same value tf.GradientTape.jacobian returns per row (model/tensorflow.py:58-62) restricted
        to the t==t' blocks.
        """
        acts, zs = self._acts(xi, with_z=True)
        f = acts[-1]
        R = f.shape[0]
        L = len(self.W)
        # cot[r, k, j] : d f_k / d z_l[j] walking back through the layers
        if self.act[-1] == "linear":
            cot = np.broadcast_to(self.W[-1].T[None, :, :], (R, self.n_out, self.W[-1].shape[0])).copy()
        else:
            cot = act_s1(self.act[-1], zs[-1], f)[:, :, None] * self.W[-1].T[None, :, :]
        for l in range(L - 2, -1, -1):
            if self.act[l] != "linear":
                cot = cot * act_s1(self.act[l], zs[l], acts[l + 1])[:, None, :]
            cot = cot @ self.W[l].T
        return f, cot

    def forward_jac_hess(self, xi):
        """(R, n_in) -> f, J (R,n_out,n_in), Hs (R,n_out,n_in,n_in) (second derivatives).

        Forward second-order propagation
            D_l = diag(s'(z_l)) W_l^T D_{l-1}
            S_l[i,p,q] = s''(z_l[i]) P[i,p] P[i,q] + s'(z_l[i]) (W_l^T S_{l-1})[i,p,q],  P = W_l^T D_{l-1}
        with s' = d1(a), s'' = r2(a) d1(a) (table above; tanh: 1-a^2, -2a(1-a^2)).  Value-equivalent to the
        tf.hessians / jax.hessian call of model/tensorflow.py:80-85, model/jax.py:74 per row.
        """
        xi = np.asarray(xi, dtype=np.float64)
        R, nin = xi.shape
        a = xi
        D = np.broadcast_to(np.eye(nin)[None], (R, nin, nin)).copy()
        S = np.zeros((R, nin, nin, nin))
        for w, b, name in zip(self.W, self.b, self.act):
            P = np.einsum("ji,rjp->rip", w, D)
            WS = np.einsum("ji,rjpq->ripq", w, S)
            z = a @ w + b
            if name != "linear":
                a = act_f(name, z)
                s1 = act_s1(name, z, a)
                s2 = act_s2(name, z, a)
                D = s1[:, :, None] * P
                S = s2[:, :, None, None] * P[:, :, :, None] * P[:, :, None, :] + s1[:, :, None, None] * WS
            else:
                a, D, S = z, P, WS
        return a, D, S


# --------------------------------------------------------------------------------------
# One-step map Phi(x_prev, u) and its per-row derivatives (tiles), for the three integrators
# --------------------------------------------------------------------------------------
def step_rows(net: MLP, kind: int, DT: float, x_prev, u, want_hess=False, extra=None):
    """Rows (R,nx),(R,nu) -> Phi (R,nx), dPhi (R,nx,nx+nu) [, d2Phi (R,nx,nx+nu,nx+nu)].

    DISCRET: Phi = x + f(x,u)          integrator/discret.py:27
    UNITY  : Phi = f(x,u)              integrator/unity.py:29
    RK4    : Phi = x + DT/6 (k1+2k2+2k3+k4), u held over the step      integrator/rk4.py:69-80
             stage Jacobians chained as dk_{i+1} = Jf(xi_i) (I + c_i DT [dk_i; 0])   rk4.py:147-159
             stage Hessians   h_{i+1} = R^T Hf(xi_i) R + c_i DT sum_j Jf[:,j] h_i[j]   rk4.py:246-266

    NOTE the reference adds the identity d x/d x of DISCRET and RK4 *outside* the model
    Jacobian (discret.py:52, rk4.py:168); here it is folded into dPhi, the dense assembly
    below therefore adds no extra +I.  Same numbers.
    """
    x_prev = np.asarray(x_prev, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    R, nx = x_prev.shape
    nu = u.shape[1]
    nin = nx + nu
    Ex = np.zeros((nx, nin))
    Ex[:, :nx] = np.eye(nx)

    # extra (R, ne): per-row network inputs after [x | u] -- [tvp_t ; p] in the reference's concatenation order
    # (model/tensorflow.py:39-47); they get no derivative columns (tensorflow.py:65-66)
    def net_eval(xs):
        xi = np.concatenate([xs, u] + ([np.asarray(extra, dtype=np.float64)] if extra is not None else []), axis=1)
        if want_hess:
            f, J, S = net.forward_jac_hess(xi)
            return f, J[:, :, :nin], S[:, :, :nin, :nin]
        f, J = net.forward_jac(xi)
        return f, J[:, :, :nin], None

    if kind in (DISCRET, UNITY):
        f, J, Hs = net_eval(x_prev)
        if kind == DISCRET:
            return x_prev + f, J + Ex[None], Hs
        return f, J, Hs

    assert kind == RK4
    cs = (0.5, 0.5, 1.0)
    wts = (1.0, 2.0, 2.0, 1.0)
    k, dk, hk = net_eval(x_prev)
    acc_k = wts[0] * k
    acc_dk = wts[0] * dk
    acc_h = wts[0] * hk if want_hess else None
    for s in range(3):
        c = cs[s] * DT
        Rm = np.broadcast_to(np.eye(nin)[None], (R, nin, nin)).copy()
        Rm[:, :nx, :] += c * dk
        kn, Jn, Hn = net_eval(x_prev + c * k)
        dkn = Jn @ Rm
        if want_hess:
            hn = np.einsum("rpa,rkab,rbq->rkpq", Rm.transpose(0, 2, 1), Hn, Rm)
            hn = hn + c * np.einsum("rkj,rjpq->rkpq", Jn[:, :, :nx], hk)
            hk = hn
            acc_h = acc_h + wts[s + 1] * hn
        k, dk = kn, dkn
        acc_k = acc_k + wts[s + 1] * kn
        acc_dk = acc_dk + wts[s + 1] * dkn
    phi = x_prev + (DT / 6.0) * acc_k
    dphi = (DT / 6.0) * acc_dk + Ex[None]
    d2phi = (DT / 6.0) * acc_h if want_hess else None
    return phi, dphi, d2phi


def rolling_rows(x_prev, u, hist_x, hist_u, window, forward=True):
    """Window inputs of a rolling model: (H,nx),(H,nu) + history (w-1,nx),(w-1,nu) -> (H, w*nx), (H, w*nu).

    model/tensorflow.py:112-130 (rolling_input) == _gather_input_V2 (tensorflow.py:205-233) and
    model/jax.py:137-151 (_gather_input + _slide_input): row t = rows t..t+w-1 of [history ; x_prev]
    flattened oldest first (forward) or newest first (forward_rolling=False)."""
    xe = np.concatenate([np.asarray(hist_x, dtype=np.float64).reshape(window - 1, -1), x_prev], axis=0)
    ue = np.concatenate([np.asarray(hist_u, dtype=np.float64).reshape(window - 1, -1), u], axis=0)
    H = x_prev.shape[0]
    order = slice(None) if forward else slice(None, None, -1)
    xr = np.stack([xe[t:t + window][order].reshape(-1) for t in range(H)], axis=0)
    ur = np.stack([ue[t:t + window][order].reshape(-1) for t in range(H)], axis=0)
    return xr, ur


def step_rows_rolling(net: MLP, kind: int, x_prev, u, hist_x, hist_u, window, forward=True, want_hess=False,
                      extra=None):
    """Rolling-window rows: Phi (H,nx), dPhi (H,nx,w*(nx+nu)) [, d2Phi (H,nx,w*(nx+nu),w*(nx+nu))] with the
    derivative columns in the network's input order [x window | u window].  DISCRET adds x_{t-1}
    (integrator/discret.py:27) and its identity at the window slot of the current state; UNITY adds nothing.
    The reference has no RK4 for rolling models."""
    assert kind in (DISCRET, UNITY)
    x_prev = np.asarray(x_prev, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    H, nx = x_prev.shape
    nu = u.shape[1]
    tw = window * (nx + nu)
    xr, ur = rolling_rows(x_prev, u, hist_x, hist_u, window, forward)
    xi = np.concatenate([xr, ur] + ([np.asarray(extra, dtype=np.float64)] if extra is not None else []), axis=1)
    if want_hess:
        f, J, S = net.forward_jac_hess(xi)
        J, S = J[:, :, :tw], S[:, :, :tw, :tw]
    else:
        f, J = net.forward_jac(xi)
        J, S = J[:, :, :tw], None
    if kind == DISCRET:
        cur = (window - 1) * nx if forward else 0
        J = J.copy()
        J[:, np.arange(nx), cur + np.arange(nx)] += 1.0
        return x_prev + f, J, S
    return f, J, S


# --------------------------------------------------------------------------------------
# Problem description shared by all callbacks
# --------------------------------------------------------------------------------------
class Problem:
    """One NLP family: dims, integrator, network, quadratic objective, optional box rows.

    Objective (covers the two instances in the reference, examples/lotka_volterra/run.py:79-84
    linear  sum(u*c)  and test.py:55-60 quadratic sum((u-2)^2)):
        f = sum_t (x_t-xref_t)^T Q (x_t-xref_t) + (u_t-uref_t)^T R (u_t-uref_t) + cx_t.x_t + cu_t.u_t
    Box rows (BASELINE config 5): extra rows  g_box = states.ravel()  with bounds [lo, hi]
    (an INTER-type Constraint in the sense of constraints.py:57-63).
    """

    def __init__(self, net, H, nx, nu, kind=DISCRET, DT=1.0, Q=None, R=None, xref=None, uref=None,
                 cx=None, cu=None, box=None, extra=None, window=1, forward_rolling=True, hist_x=None, hist_u=None,
                 QT=None):
        self.extra = None if extra is None else np.asarray(extra, dtype=np.float64).reshape(H, -1)
        # rolling-window models (model/tensorflow.py:132, model/jax.py:93): tile width tw = window*(nx+nu)
        self.window, self.forward_rolling = int(window), bool(forward_rolling)
        self.tw = self.window * (nx + nu)
        self.hist_x = None if hist_x is None else np.asarray(hist_x, dtype=np.float64).reshape(self.window - 1, nx)
        self.hist_u = None if hist_u is None else np.asarray(hist_u, dtype=np.float64).reshape(self.window - 1, nu)
        assert self.window == 1 or (kind != RK4 and self.hist_x is not None and self.hist_u is not None)
        assert net.n_in == self.tw + (0 if self.extra is None else self.extra.shape[1]) and net.n_out == nx
        self.net, self.H, self.nx, self.nu, self.kind, self.DT = net, H, nx, nu, kind, float(DT)
        self.n = H * (nx + nu)
        self.Q = np.eye(nx) if Q is None else np.asarray(Q, dtype=np.float64).reshape(nx, nx)
        self.R = 0.1 * np.eye(nu) if R is None else np.asarray(R, dtype=np.float64).reshape(nu, nu)
        # terminal cost: the last step's state weight (a member of the same family; None = Q)
        self.QT = self.Q if QT is None else np.asarray(QT, dtype=np.float64).reshape(nx, nx)

        def _tv(v, d):
            if v is None:
                return np.zeros((H, d))
            return np.broadcast_to(np.asarray(v, dtype=np.float64), (H, d)).copy()

        self.xref, self.uref = _tv(xref, nx), _tv(uref, nu)
        self.cx, self.cu = _tv(cx, nx), _tv(cu, nu)
        self.box = None
        if box is not None:
            lo, hi = box
            self.box = (np.broadcast_to(np.asarray(lo, dtype=np.float64), (nx,)).copy(),
                        np.broadcast_to(np.asarray(hi, dtype=np.float64), (nx,)).copy())
        self.m = H * nx + (H * nx if self.box is not None else 0)

    # ---- optimizer/ipopt.py:20-28
    def split(self, z):
        H, nx, nu = self.H, self.nx, self.nu
        z = np.asarray(z, dtype=np.float64)
        return z[: H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)

    # ---- optimizer/ipopt.py:30-42 with the quadratic family standing in for objective/jax.py
    def objective(self, z):
        x, u = self.split(z)
        dx, du = x - self.xref, u - self.uref
        return float(np.einsum("ti,ij,tj->", dx[:-1], self.Q, dx[:-1]) + dx[-1] @ self.QT @ dx[-1]
                     + np.einsum("ti,ij,tj->", du, self.R, du) + np.sum(self.cx * x) + np.sum(self.cu * u))

    def gradient(self, z):
        x, u = self.split(z)
        gx = (x - self.xref) @ (self.Q + self.Q.T).T + self.cx
        gx[-1] = (x[-1] - self.xref[-1]) @ (self.QT + self.QT.T).T + self.cx[-1]
        gu = (u - self.uref) @ (self.R + self.R.T).T + self.cu
        return np.concatenate([gx.ravel(), gu.ravel()])

    def objective_hessian(self):
        n, H, nx, nu = self.n, self.H, self.nx, self.nu
        Hm = np.zeros((n, n))
        for t in range(H):
            Qt = self.QT if t == H - 1 else self.Q
            Hm[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = Qt + Qt.T
            o = H * nx + t * nu
            Hm[o:o + nu, o:o + nu] = self.R + self.R.T
        return Hm

    # ---- per-row tiles
    def tiles(self, z, x0, want_hess=False):
        x, u = self.split(z)
        x_prev = np.concatenate([np.asarray(x0, dtype=np.float64).reshape(1, -1), x[:-1]], axis=0)
        if self.window > 1:
            phi, dphi, d2phi = step_rows_rolling(self.net, self.kind, x_prev, u, self.hist_x, self.hist_u, self.window,
                                                 self.forward_rolling, want_hess, extra=self.extra)
        else:
            phi, dphi, d2phi = step_rows(self.net, self.kind, self.DT, x_prev, u, want_hess, extra=self.extra)
        return x, phi, dphi, d2phi

    def tile_columns(self, t):
        """Index into z of every tile / block column of step t, -1 where the window slot is data (x0, history).
        Window slot j of the states is row t+j (forward) of [history ; x0 ; states[:-1]]: x0 and the history have
        no column (integrator/discret.py:52-53 drops the first state block), state block s = that row - w."""
        H, nx, nu, w = self.H, self.nx, self.nu, self.window
        cols = -np.ones(self.tw, dtype=np.int64)
        for j in range(w):
            tau = t + (j - (w - 1) if self.forward_rolling else -j)   # index into [x0 ; states]
            if tau >= 1:
                cols[j * nx:(j + 1) * nx] = (tau - 1) * nx + np.arange(nx)
            if tau >= 0:
                cols[w * nx + j * nu:w * nx + (j + 1) * nu] = H * nx + tau * nu + np.arange(nu)
        return cols

    # ---- optimizer/ipopt.py:44-52 ; integrator/discret.py:13-30
    def constraints(self, z, x0):
        x, phi, _, _ = self.tiles(z, x0)
        g = (phi - x).ravel()
        if self.box is not None:
            g = np.concatenate([g, x.ravel()])
        return g

    # ---- optimizer/ipopt.py:88-96 ; integrator/discret.py:32-58 ; rk4.py:113-178
    def jacobian(self, z, x0):
        H, nx, nu, n = self.H, self.nx, self.nu, self.n
        _, _, dphi, _ = self.tiles(z, x0)
        J = np.zeros((self.m, n))
        for t in range(H):
            r = slice(t * nx, (t + 1) * nx)
            J[r, t * nx:(t + 1) * nx] -= np.eye(nx)
            cols = self.tile_columns(t)     # x0 / history are data, not variables: no column
            keep = cols >= 0
            J[r, cols[keep]] += dphi[t][:, keep]
        if self.box is not None:
            J[H * nx:, :H * nx] = np.eye(H * nx)
        return J

    def tiles_AB(self, z, x0):
        """Compact contract: g (H*nx,), A (H,nx,nx) = dPhi/dx_prev, Bt (H,nx,nu) = dPhi/du."""
        assert self.window == 1
        x, phi, dphi, _ = self.tiles(z, x0)
        return (phi - x).ravel(), dphi[:, :, : self.nx].copy(), dphi[:, :, self.nx:].copy()

    def constraint_bounds(self):
        """optimizer/ipopt.py:104-108 ; integrator/base.py:119-123"""
        cl = np.zeros(self.H * self.nx)
        cu = np.zeros(self.H * self.nx)
        if self.box is not None:
            cl = np.concatenate([cl, np.tile(self.box[0], self.H)])
            cu = np.concatenate([cu, np.tile(self.box[1], self.H)])
        return cl, cu

    # ---- optimizer/ipopt.py:66-86 (dense Lagrangian Hessian before the tril gather)
    def lagrangian_hessian(self, z, x0, lam, sigma):
        """sigma * d2f + sum_i lam_i d2g_i, dense (n,n).  Box rows are linear: no contribution.

        Placement of the per-row (nx+nu)^2 blocks follows integrator/discret.py:61-81 /
        rk4.py:267-285: row t's x-part maps to variable block t-1 (dropped for t=0), its
        u-part to control block t.
        """
        H, nx, nu, n = self.H, self.nx, self.nu, self.n
        lam = np.asarray(lam, dtype=np.float64)
        _, _, _, d2phi = self.tiles(z, x0, want_hess=True)
        Hm = sigma * self.objective_hessian()
        for t in range(H):
            blk = np.einsum("k,kpq->pq", lam[t * nx:(t + 1) * nx], d2phi[t])
            cols = self.tile_columns(t)
            keep = np.nonzero(cols >= 0)[0]
            Hm[np.ix_(cols[keep], cols[keep])] += blk[np.ix_(keep, keep)]
        return Hm

    def gauss_newton_hessian(self, z, x0, w, sigma):
        """sigma * d2f + sum_t T_t^T diag(w_t) T_t, dense (n,n), T_t = d Phi / d[x_{t-1} | u_t] the row's Jacobian tile.

        No reference counterpart: the reference's own Hessian callback is the EXACT Lagrangian Hessian above
        (optimizer/ipopt.py:66-86); BASELINE.json's north_star names the Gauss-Newton variant -- the first-order model
        of how a weight w_t on step t's successor state curves the problem in (x_{t-1}, u_t), i.e. what remains of
        lagrangian_hessian when the network's second derivatives are dropped and the multipliers are replaced by a
        positive weighting.  Same block placement, hence the same sparsity pattern, as the exact one."""
        H, nx, n = self.H, self.nx, self.n
        w = np.ones(H * nx) if w is None else np.asarray(w, dtype=np.float64)
        J = self.tiles(z, x0)[2]
        Hm = sigma * self.objective_hessian()
        for t in range(H):
            blk = np.einsum("k,kp,kq->pq", w[t * nx:(t + 1) * nx], J[t], J[t])
            cols = self.tile_columns(t)
            keep = np.nonzero(cols >= 0)[0]
            Hm[np.ix_(cols[keep], cols[keep])] += blk[np.ix_(keep, keep)]
        return Hm

    def gauss_newton_values(self, z, x0, w, sigma):
        r, c = self.hessian_structure()
        return self.gauss_newton_hessian(z, x0, w, sigma)[r, c]

    def hessian_structure(self):
        """Exact structural lower-triangular pattern, row-major ordered like
        np.nonzero(np.tril(.)) in optimizer/ipopt.py:55-62.  (The reference samples three random
        points, integrator/base.py:89-115; the structural pattern is its superset.)"""
        H, nx, nu, n = self.H, self.nx, self.nu, self.n
        M = np.zeros((n, n), dtype=bool)
        for t in range(H):
            uo = H * nx + t * nu
            M[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = True   # objective Q block
            M[uo:uo + nu, uo:uo + nu] = True                     # objective R block
            cols = self.tile_columns(t)
            cols = cols[cols >= 0]
            M[np.ix_(cols, cols)] = True                         # every pair of variables step t reads
        return np.nonzero(np.tril(M))

    def hessian_values(self, z, x0, lam, sigma):
        r, c = self.hessian_structure()
        return self.lagrangian_hessian(z, x0, lam, sigma)[r, c]

    # ---- batched convenience (python loop over problems = "reference-shaped" CPU mode)
    def eval_batch(self, Z, X0, dense=True):
        B = Z.shape[0]
        f = np.empty(B)
        grad = np.empty((B, self.n))
        g = np.empty((B, self.m))
        jac = np.empty((B, self.m, self.n)) if dense else None
        for b in range(B):
            f[b] = self.objective(Z[b])
            grad[b] = self.gradient(Z[b])
            g[b] = self.constraints(Z[b], X0[b])
            if dense:
                jac[b] = self.jacobian(Z[b], X0[b])
        return f, grad, g, jac


# --------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md 8(d) -- shared by tests, smoke and bench
# --------------------------------------------------------------------------------------
def synthetic_inputs(B, H, nx, nu, seed=1):
    rng = np.random.default_rng(seed)
    X0 = rng.uniform(-1.0, 1.0, size=(B, nx))
    states = rng.normal(0.0, 1.0, size=(B, H, nx))
    u = rng.uniform(-1.0, 1.0, size=(B, H, nu))
    Z = np.concatenate([states.reshape(B, -1), u.reshape(B, -1)], axis=1)
    return Z, X0


def warm_start_shift(prev, H, nx, nu):
    """optimizer/ipopt.py:141-147 == slsqp.py:155-161: drop step 0, repeat the last step."""
    prev = np.asarray(prev, dtype=np.float64)
    return np.concatenate([prev[nx:nx * H], prev[nx * (H - 1):nx * H],
                           prev[nx * H + nu:(nx + nu) * H], prev[nx * H + nu * (H - 1):nx * H + nu * H]])


def cold_start(x0, H, nu):
    """optimizer/ipopt.py:149: [x0 tiled H ; zeros(H*nu)]."""
    return np.concatenate([np.tile(np.asarray(x0, dtype=np.float64), H), np.zeros(H * nu)])


def domain_bounds(states_constraint, control_constraint, H):
    """constraints.py:26-30."""
    lb = [e[0] for e in states_constraint] * H + [e[0] for e in control_constraint] * H
    ub = [e[1] for e in states_constraint] * H + [e[1] for e in control_constraint] * H
    return np.array(lb, dtype=np.float64), np.array(ub, dtype=np.float64)
