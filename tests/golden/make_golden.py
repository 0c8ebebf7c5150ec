#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own NumPy code.

Run only in the build container (needs /root/reference, read-only; it never travels to the GPU
box -- only the .npz files this script writes do):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

How the reference is imported: its package __init__ files pull tensorflow / jax,
which are not installed, so empty package shells are registered in sys.modules first and the
NumPy-only modules are imported from their files: model/base.py, integrator/{base,discret,unity,
rk4}.py, constraints.py, objective/base.py, optimizer/{base,slsqp,ipopt}.py, controller.py.
optimizer/ipopt.py has a top-level ``import cyipopt``; an attribute-less placeholder module is
registered for that name so the pure-NumPy glue class ``IpoptProblem`` can be imported.
``Ipopt.solve`` (the only user of cyipopt) is never called.

What the reference cannot supply: Model.jacobian / Model.hessian / ObjectiveFunc.gradient are
TensorFlow / JAX autodiff calls there.  The plug-ins below subclass the reference's own ``Model``
and ``ObjectiveFunc`` ABCs, produce the reference's *layouts* (block layout [all-x | all-u],
model/tensorflow.py:68-73) and take the per-row derivative *values* from oracle.MLP, which is
pinned separately by torch.func AD and finite differences (tests/test_oracle.py).
Everything downstream of those plug-ins in the stored outputs is reference arithmetic.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pyNeuralEMPC"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import nempc_oracle as orc  # noqa: E402


def _shell(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    root = _shell("pyNeuralEMPC", REF)
    for sub in ("model", "integrator", "optimizer", "objective"):
        setattr(root, sub, _shell(f"pyNeuralEMPC.{sub}", f"{REF}/{sub}"))
    sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))  # placeholder, see docstring
    ref = types.SimpleNamespace()
    ref.model_base = _load("pyNeuralEMPC.model.base", f"{REF}/model/base.py")
    ref.constraints = _load("pyNeuralEMPC.constraints", f"{REF}/constraints.py")
    ref.objective_base = _load("pyNeuralEMPC.objective.base", f"{REF}/objective/base.py")
    ref.integ_base = _load("pyNeuralEMPC.integrator.base", f"{REF}/integrator/base.py")
    ref.discret = _load("pyNeuralEMPC.integrator.discret", f"{REF}/integrator/discret.py")
    ref.unity = _load("pyNeuralEMPC.integrator.unity", f"{REF}/integrator/unity.py")
    ref.rk4 = _load("pyNeuralEMPC.integrator.rk4", f"{REF}/integrator/rk4.py")
    ref.opt_base = _load("pyNeuralEMPC.optimizer.base", f"{REF}/optimizer/base.py")
    ref.slsqp = _load("pyNeuralEMPC.optimizer.slsqp", f"{REF}/optimizer/slsqp.py")
    ref.ipopt = _load("pyNeuralEMPC.optimizer.ipopt", f"{REF}/optimizer/ipopt.py")
    opt = sys.modules["pyNeuralEMPC.optimizer"]
    opt.Ipopt, opt.Optimizer, opt.Slsqp = ref.ipopt.Ipopt, ref.opt_base.Optimizer, ref.slsqp.Slsqp
    ref.controller = _load("pyNeuralEMPC.controller", f"{REF}/controller.py")
    # model/jax.py: its window -> variable projection gen_jac_proj_mat (jax.py:8-20) is plain NumPy; the module's
    # top-level `import jax` / `import jax.numpy` only bind names used inside the JAX model classes, so attribute-less
    # placeholders (as for cyipopt above) are enough to import it.  Nothing of JAX is called.
    if "jax" not in sys.modules:
        jax_ph = types.ModuleType("jax")
        jax_ph.numpy = types.ModuleType("jax.numpy")
        sys.modules["jax"], sys.modules["jax.numpy"] = jax_ph, jax_ph.numpy
    ref.model_jax = _load("pyNeuralEMPC.model.jax", f"{REF}/model/jax.py")
    return ref


def make_plugins(ref):
    class NumpyMLPModel(ref.model_base.Model):
        """fp64 stand-in for KerasTFModel: same call signature and output layouts."""

        def __init__(self, net, x_dim, u_dim, p_dim=0, tvp_dim=0):
            super().__init__(x_dim, u_dim, p_dim, tvp_dim)
            self.net = net

        def _gather(self, x, u, p, tvp):
            # KerasTFModel._gather_input (model/tensorflow.py:39-47): [x | u | tvp | p]; the reference's own p
            # branch builds a 3-D array and cannot run, the intended "repeat p on every row" is used
            parts = [x, u]
            if tvp is not None:
                parts.append(tvp)
            if p is not None:
                parts.append(np.tile(np.asarray(p).reshape(1, -1), (x.shape[0], 1)))
            return np.concatenate(parts, axis=1)

        def forward(self, x, u, p=None, tvp=None):
            return self.net.forward(self._gather(x, u, p, tvp))

        def jacobian(self, x, u, p=None, tvp=None):
            H, nx, nu = x.shape[0], self.x_dim, self.u_dim
            _, J = self.net.forward_jac(self._gather(x, u, p, tvp))   # extra columns dropped below (tensorflow.py:65-66)
            out = np.zeros((H * nx, H * nx + H * nu))
            for t in range(H):
                out[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = J[t, :, :nx]
                out[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu] = J[t, :, nx:nx + nu]
            return out

        def hessian(self, x, u, p=None, tvp=None):
            H, nx, nu = x.shape[0], self.x_dim, self.u_dim
            _, _, S = self.net.forward_jac_hess(self._gather(x, u, p, tvp))
            n = H * (nx + nu)
            out = np.zeros((H, nx, n, n))
            for t in range(H):
                xs = slice(t * nx, (t + 1) * nx)
                us = slice(H * nx + t * nu, H * nx + (t + 1) * nu)
                out[t][:, xs, xs] = S[t][:, :nx, :nx]
                out[t][:, xs, us] = S[t][:, :nx, nx:nx + nu]
                out[t][:, us, xs] = S[t][:, nx:nx + nu, :nx]
                out[t][:, us, us] = S[t][:, nx:nx + nu, nx:nx + nu]
            return out

    class NumpyRollingMLPModel(ref.model_base.Model):
        """fp64 stand-in for KerasTFModelRollingInput (model/tensorflow.py:132-340) /
        DiffDiscretJaxModelRollingWindow (model/jax.py:93-259): same constructor arguments, set_prev_data, and
        output layouts ([all x_t_1 | all u] columns).  The window -> variable projection is written the way
        gen_jac_proj_mat (model/jax.py:8-21) writes it: variable (step i, component k) sits in the window of
        step o at slot (w-1) - (o-i) (forward) or o-i (forward_rolling=False) for i <= o <= i+w-1."""

        def __init__(self, net, x_dim, u_dim, p_dim=0, tvp_dim=0, rolling_window=2, forward_rolling=True):
            super().__init__(x_dim, u_dim, p_dim, tvp_dim)
            self.net, self.rolling_window, self.forward_rolling = net, rolling_window, forward_rolling
            self.prev_x = self.prev_u = self.prev_tvp = None

        def set_prev_data(self, x_prev, u_prev, tvp_prev=None):
            assert x_prev.shape == (self.rolling_window - 1, self.x_dim)
            assert u_prev.shape == (self.rolling_window - 1, self.u_dim)
            self.prev_x, self.prev_u, self.prev_tvp = x_prev, u_prev, tvp_prev

        def _roll(self, prev, cur):
            ext = np.concatenate([prev, cur], axis=0)
            w = self.rolling_window
            rows = [ext[i:i + w] for i in range(cur.shape[0])]
            if not self.forward_rolling:
                rows = [r[::-1] for r in rows]
            return np.stack([r.reshape(-1) for r in rows], axis=0)

        def _gather(self, x, u, p, tvp):   # _gather_input_V2, tensorflow.py:205-233
            parts = [self._roll(self.prev_x, x), self._roll(self.prev_u, u)]
            if tvp is not None:
                parts.append(self._roll(self.prev_tvp, tvp))
            if p is not None:
                parts.append(np.tile(np.asarray(p).reshape(1, -1), (x.shape[0], 1)))
            return np.concatenate(parts, axis=1)

        def _proj_written(self, T, dim):
            # (T*dim variables, T steps, w*dim window slots) 0/1 selector, this file's reading of gen_jac_proj_mat,
            # extended to forward_rolling=False (which the reference leaves as a TODO, jax.py:186)
            w = self.rolling_window
            P = np.zeros((T * dim, T, w * dim))
            for i in range(T):
                for k in range(dim):
                    for o in range(i, min(T, i + w)):
                        slot = (w - 1) - (o - i) if self.forward_rolling else (o - i)
                        P[i * dim + k, o, slot * dim + k] = 1.0
            return P

        def _proj(self, T, dim):
            """Forward order: the reference's OWN gen_jac_proj_mat (model/jax.py:8-20), imported and called here, so the
            rolling fixtures pin the band against that function itself; the hand-written form must agree with it.
            Reverse order (no reference implementation): the hand-written form."""
            if not self.forward_rolling:
                return self._proj_written(T, dim)
            P = ref.model_jax.gen_jac_proj_mat(T, dim, self.rolling_window).reshape(T * dim, T, self.rolling_window * dim)
            assert np.array_equal(P, self._proj_written(T, dim)), "hand-written projection != gen_jac_proj_mat"
            return P

        def forward(self, x, u, p=None, tvp=None):
            return self.net.forward(self._gather(x, u, p, tvp))

        def jacobian(self, x, u, p=None, tvp=None):
            H, nx, nu, w = x.shape[0], self.x_dim, self.u_dim, self.rolling_window
            _, J = self.net.forward_jac(self._gather(x, u, p, tvp))
            jx = np.einsum("okc,voc->okv", J[:, :, :w * nx], self._proj(H, nx))
            ju = np.einsum("okc,voc->okv", J[:, :, w * nx:w * (nx + nu)], self._proj(H, nu))
            return np.concatenate([jx, ju], axis=2).reshape(H * nx, H * (nx + nu))

        def hessian(self, x, u, p=None, tvp=None):
            H, nx, nu, w = x.shape[0], self.x_dim, self.u_dim, self.rolling_window
            _, _, S = self.net.forward_jac_hess(self._gather(x, u, p, tvp))
            tw = w * (nx + nu)
            P = np.zeros((H * (nx + nu), H, tw))            # all variables [x | u] x steps x window columns
            P[:H * nx, :, :w * nx] = self._proj(H, nx)
            P[H * nx:, :, w * nx:] = self._proj(H, nu)
            return np.einsum("aoc,okcd,bod->okab", P, S[:, :, :tw, :tw], P)

    class QuadObjective(ref.objective_base.ObjectiveFunc):
        """Closed-form member of the objective family; call-site signature of ipopt.py:33,40,71."""

        def __init__(self, prob):
            super().__init__()
            self.prob = prob

        def _z(self, states, u):
            return np.concatenate([states.ravel(), u.ravel()])

        def forward(self, states, u, p=None, tvp=None):
            return self.prob.objective(self._z(states, u))

        def gradient(self, states, u, p=None, tvp=None):
            return self.prob.gradient(self._z(states, u))

        def hessian(self, states, u, p=None, tvp=None):
            return self.prob.objective_hessian()

        def hessianstructure(self, H, model):
            return (self.prob.objective_hessian() != 0.0).astype(np.float64)

    class BoxStateRows(ref.constraints.Constraint):
        """rows = states.ravel() in [lo, hi]  (no concrete row constraint exists in the reference)."""

        def __init__(self, lo, hi, nx, nu):
            self.lo, self.hi, self.nx, self.nu = np.asarray(lo, float), np.asarray(hi, float), nx, nu

        def forward(self, x, u, p=None, tvp=None):
            return x.reshape(-1).copy()

        def jacobian(self, x, u, p=None, tvp=None):
            H = x.shape[0]
            return np.concatenate([np.eye(H * self.nx), np.zeros((H * self.nx, H * self.nu))], axis=1)

        def hessian(self, x, u, p=None, tvp=None):
            H = x.shape[0]
            n = H * (self.nx + self.nu)
            return np.zeros((H * self.nx, n, n))

        def get_dim(self, H):
            return H * self.nx

        def get_lower_bounds(self, H):
            return np.tile(self.lo, H)

        def get_upper_bounds(self, H):
            return np.tile(self.hi, H)

    return NumpyMLPModel, QuadObjective, BoxStateRows, NumpyRollingMLPModel


CASES = {
    # name: (nx, nu, hidden, H, kind, DT, box, B, with_hessian)
    "c1_discret": (2, 1, [30, 30], 10, orc.DISCRET, 1.0, None, 3, True),
    "c2_discret": (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 4, True),
    "c2_unity": (2, 1, [64, 64], 20, orc.UNITY, 1.0, None, 2, True),
    "c2_rk4": (2, 1, [64, 64], 20, orc.RK4, 0.1, None, 2, True),     # nx+nu=3: rk4.hessian usable
    "c3_rk4": (6, 3, [128, 128, 128], 30, orc.RK4, 0.1, None, 2, False),
    "c3_discret": (6, 3, [128, 128, 128], 30, orc.DISCRET, 1.0, None, 1, False),
    "c5_box": (2, 1, [64, 64], 50, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True),
    "odd_dims": (3, 2, [48, 32], 7, orc.RK4, 0.05, None, 2, False),  # ragged: widths not equal, H odd
    "h1": (2, 1, [16], 1, orc.DISCRET, 1.0, None, 2, True),            # degenerate horizon
    # time-varying + constant parameters as extra network inputs: (.., with_hessian, p_dim, tvp_dim)
    "tvp_p_discret": (2, 1, [32, 32], 6, orc.DISCRET, 1.0, None, 2, True, 1, 2),
    "tvp_p_rk4": (2, 1, [32, 32], 6, orc.RK4, 0.2, None, 2, True, 1, 2),
    # activation family (ABI v6): (.., with_hessian, p_dim, tvp_dim, activations) -- one configs[1]-shaped case (2/1, 2x64,
    # H=20, Discret, with the Lagrangian Hessian) and one configs[2]-shaped case (6/3, 3x128, RK4; a short horizon keeps the
    # fixture small) per activation, through the reference's own integrators / IpoptProblem like every case above
    "act_relu_c2": (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 2, True, 0, 0, "relu"),
    "act_sigmoid_c2": (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 2, True, 0, 0, "sigmoid"),
    "act_softplus_c2": (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 2, True, 0, 0, "softplus"),
    "act_elu_c2": (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 2, True, 0, 0, "elu"),
    "act_relu_c3": (6, 3, [128, 128, 128], 6, orc.RK4, 0.1, None, 1, False, 0, 0, "relu"),
    "act_sigmoid_c3": (6, 3, [128, 128, 128], 6, orc.RK4, 0.1, None, 1, False, 0, 0, "sigmoid"),
    "act_softplus_c3": (6, 3, [128, 128, 128], 6, orc.RK4, 0.1, None, 1, False, 0, 0, "softplus"),
    "act_elu_c3": (6, 3, [128, 128, 128], 6, orc.RK4, 0.1, None, 1, False, 0, 0, "elu"),
    # a different activation on every layer, the output layer included (generic kernel): box rows + Hessian under Discret,
    # and RK4 with nx+nu = 3 so that the reference's own rk4.hessian applies
    "act_mixed_box": (2, 1, [32, 24, 16], 9, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True, 0, 0,
                      ["relu", "softplus", "elu", "sigmoid"]),
    "act_mixed_rk4": (2, 1, [24, 24], 7, orc.RK4, 0.2, None, 2, True, 0, 0, ["elu", "sigmoid", "tanh"]),
    "act_linear_hidden": (2, 1, [16, 16], 5, orc.UNITY, 1.0, None, 2, True, 0, 0, ["linear", "tanh", "linear"]),
    # networks outside the register-resident kernels (round 4: the layer-at-a-time GEMM path, csrc/kernels_layered.hip):
    # a width beyond 128, more than three hidden layers, and a deep ragged mix under RK4 (short horizons keep the files small)
    "wide256_c2": (2, 1, [256, 256], 6, orc.DISCRET, 1.0, None, 2, True),
    "deep4_c2": (2, 1, [64, 64, 64, 64], 6, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True),
    "deep5_mixed_rk4": (2, 1, [144, 96, 96, 40, 24], 5, orc.RK4, 0.1, None, 2, True, 0, 0,
                        ["tanh", "relu", "tanh", "softplus", "elu", "linear"]),
    # the parameterised / remaining monotone activations (ABI v7): leaky_relu(alpha), selu, elu(alpha != 1)
    "act_param_box": (2, 1, [32, 24, 16], 9, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True, 0, 0,
                      ["leaky_relu:0.1", "selu", "elu:0.5", "leaky_relu"]),
    "act_selu_rk4": (2, 1, [48, 48], 7, orc.RK4, 0.2, None, 2, True, 0, 0, "selu"),
    # per-layer mixes with a linear output layer, width <= 128: the register-resident matrix-core kernels with run-time
    # activation codes (round 5; ActL in csrc/activations.h) -- the 3 x 128 relu / tanh / sigmoid mix on configs[1]'s dims
    # with box rows, a parameterised pair under RK4 with the Lagrangian Hessian, configs[2]'s dims under RK4, and a fourth
    # hidden layer with a mix
    "act_mix3_c2": (2, 1, [128, 128, 128], 6, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True, 0, 0, ["relu", "tanh", "sigmoid", "linear"]),
    "act_mix2_rk4": (2, 1, [64, 48], 7, orc.RK4, 0.2, None, 2, True, 0, 0, ["leaky_relu:0.1", "selu", "linear"]),
    "act_mix_c3_rk4": (6, 3, [128, 128, 128], 4, orc.RK4, 0.1, None, 1, False, 0, 0, ["tanh", "softplus", "elu:0.5", "linear"]),
    "deep4_mixed_c2": (2, 1, [64, 48, 64, 32], 6, orc.DISCRET, 1.0, None, 2, True, 0, 0,
                       ["tanh", "relu", "sigmoid", "elu", "linear"]),
    # the non-monotone activations (derivatives from the pre-activation; the layered path only)
    "act_swish_gelu_box": (2, 1, [48, 40], 9, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True, 0, 0, ["swish", "gelu", "linear"]),
    "act_gelu_rk4": (2, 1, [40, 40], 7, orc.RK4, 0.2, None, 2, True, 0, 0, "gelu"),
    # ... on the OUTPUT layer too (round 5: the layered path's output step and the generic kernel keep the pre-activation):
    # Discret with box rows, and a single hidden layer under a non-linear output layer under RK4 (the layered Hessian
    # hands that shape to the generic kernel)
    "act_zout_box": (2, 1, [40, 32], 7, orc.DISCRET, 1.0, (-2.0, 2.0), 2, True, 0, 0, ["tanh", "swish", "gelu"]),
    "act_zout_rk4": (2, 1, [24], 6, orc.RK4, 0.2, None, 2, True, 0, 0, ["gelu", "softsign"]),
}


def check_network_derivatives_by_ad(net, n_in, seed=11, rows=3, with_hess=True):
    """The one piece of every fixture that does not come from reference arithmetic -- the per-row derivative of the
    network, which the reference obtains from TensorFlow / JAX autodiff -- is cross-checked HERE, while the fixtures are
    made, against an independent autodiff (torch.func jacrev / hessian, fp64), so a fixture can never be written from
    a network derivative that only the oracle believes in."""
    import torch
    Wt = [torch.tensor(w, dtype=torch.float64) for w in net.W]
    bt = [torch.tensor(b, dtype=torch.float64) for b in net.b]

    def tact(spec):
        name, par = orc.act_split(spec)
        F = torch.nn.functional
        return {"linear": lambda z: z, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "softplus": F.softplus,
                "elu": lambda z: F.elu(z, alpha=par), "leaky_relu": lambda z: F.leaky_relu(z, negative_slope=par),
                "selu": F.selu, "swish": F.silu, "gelu": F.gelu, "softsign": F.softsign, "mish": F.mish, "exponential": torch.exp,
                "relu6": F.relu6}[name]

    def f(xi):
        a = xi
        for w, b, name in zip(Wt, bt, net.act):
            a = tact(name)(a @ w + b)
        return a

    xi = np.random.default_rng(seed).normal(size=(rows, n_in))
    fo, Jo, So = net.forward_jac_hess(xi)
    _, Jr = net.forward_jac(xi)
    for r in range(rows):
        x = torch.tensor(xi[r], dtype=torch.float64)
        Ja = torch.func.jacrev(f)(x).numpy()
        assert np.abs(Ja - Jo[r]).max() < 1e-12 and np.abs(Ja - Jr[r]).max() < 1e-12, "network Jacobian != torch AD"
        assert np.abs(f(x).numpy() - fo[r]).max() < 1e-13
        if with_hess:
            Ha = torch.func.hessian(f)(x).numpy()
            assert np.abs(Ha - So[r]).max() < 1e-11, "network Hessian != torch AD"


def build_case(ref, plugins, name, spec):
    NumpyMLPModel, QuadObjective, BoxStateRows = plugins[:3]
    nx, nu, hidden, H, kind, DT, box, B, with_h = spec[:9]
    p_dim, tvp_dim = (spec[9], spec[10]) if len(spec) > 9 else (0, 0)
    activations = spec[11] if len(spec) > 11 else None
    net = orc.MLP.random(nx + nu + p_dim + tvp_dim, hidden, nx, seed=0, activations=activations)
    check_network_derivatives_by_ad(net, nx + nu + p_dim + tvp_dim, with_hess=max(hidden) <= 64)
    rng = np.random.default_rng(7)
    xref = rng.normal(size=(H, nx)) * 0.3
    uref = rng.normal(size=(H, nu)) * 0.3
    cu = rng.normal(size=(H, nu)) * 0.1
    Q = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx))
    Rm = 0.1 * np.eye(nu)
    pvec = rng.normal(size=p_dim) if p_dim else None
    tvp = rng.normal(size=(H, tvp_dim)) if tvp_dim else None
    extra = None
    if p_dim or tvp_dim:
        extra = np.concatenate(([tvp] if tvp_dim else []) + ([np.tile(pvec.reshape(1, -1), (H, 1))] if p_dim else []), axis=1)
    prob = orc.Problem(net, H, nx, nu, kind, DT, Q=Q, R=Rm, xref=xref, uref=uref, cu=cu, box=box, extra=extra)
    model = NumpyMLPModel(net, nx, nu, p_dim, tvp_dim)
    if kind == orc.DISCRET:
        integ = ref.discret.DiscretIntegrator(model, H)
    elif kind == orc.UNITY:
        integ = ref.unity.UnityIntegrator(model, H)
    else:
        integ = ref.rk4.RK4Integrator(model, H, DT)
    ctrs = [BoxStateRows(np.full(nx, box[0]), np.full(nx, box[1]), nx, nu)] if box is not None else []
    obj = QuadObjective(prob)
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    lam = np.random.default_rng(3).normal(size=(B, prob.m))
    sigma = np.array([1.0, 0.5, 2.0, 0.0])[:B] if B <= 4 else np.ones(B)

    out = {"nx": nx, "nu": nu, "H": H, "kind": kind, "DT": DT, "hidden": np.array(hidden),
           "Z": Z, "X0": X0, "Q": Q, "R": Rm, "xref": xref, "uref": uref, "cu": cu,
           "lam": lam, "sigma": sigma, "has_box": int(box is not None), "p_dim": p_dim, "tvp_dim": tvp_dim}
    if p_dim:
        out["p"] = pvec
    if tvp_dim:
        out["tvp"] = tvp
    if box is not None:
        out["box_lo"], out["box_hi"] = np.full(nx, box[0]), np.full(nx, box[1])
    for i, (w, b) in enumerate(zip(net.W, net.b)):
        out[f"W{i}"], out[f"b{i}"] = w, b
    if activations is not None:
        out["activations"] = np.array([orc.ACT_IDS[orc.act_split(a)[0]] for a in net.act])     # NEMPC_ACT_* per layer
        if any(":" in a for a in net.act):
            out["act_param"] = np.array([orc.act_split(a)[1] for a in net.act])               # alpha of elu / leaky_relu

    f, grad, g, jac, hvals, hdense = [], [], [], [], [], []
    g_int, j_int = [], []
    for b in range(B):
        pb = ref.ipopt.IpoptProblem(X0[b], obj, ctrs, integ, p=pvec, tvp=tvp)
        f.append(pb.objective(Z[b]))
        grad.append(pb.gradient(Z[b]))
        g.append(pb.constraints(Z[b]))
        jac.append(pb.jacobian(Z[b]))
        states, u = Z[b][:H * nx].reshape(H, nx), Z[b][H * nx:].reshape(H, nu)
        g_int.append(integ.forward(states, u, X0[b], p=pvec, tvp=tvp))
        j_int.append(integ.jacobian(states, u, X0[b], p=pvec, tvp=tvp))
        if with_h:
            np.random.seed(11)  # integrator/base.py:96-97 samples with the global RNG
            rows, cols = pb.hessianstructure()
            out["h_rows"], out["h_cols"] = rows, cols
            hvals.append(pb.hessian(Z[b], lam[b], sigma[b]))
            ih = integ.hessian(states, u, X0[b], p=pvec, tvp=tvp)
            hdense.append(sigma[b] * obj.hessian(states, u) + np.einsum("i,ipq->pq", lam[b][:H * nx], ih))
    out.update(f=np.array(f), grad=np.array(grad), g=np.array(g), jac=np.array(jac),
               g_int=np.array(g_int), jac_int=np.array(j_int))
    out["cl"], out["cu_bound"] = pb.get_constraint_lower_bounds(), pb.get_constraint_upper_bounds()
    if with_h:
        out["hvals"], out["hdense"] = np.array(hvals), np.array(hdense)

    # SLSQP glue splits (slsqp.py:54-110); INTER rows crash in the reference (slsqp.py:67-68) -> eq only there
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sp = ref.slsqp.SlsqpProblem(X0[0], obj, [] if box is not None else ctrs, integ, p=pvec, tvp=tvp)
        out["slsqp_eq"] = sp.constraints(Z[0], eq=True)
        out["slsqp_eq_jac"] = sp.jacobian(Z[0], eq=True)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: n={prob.n} m={prob.m} B={B} nnz(jac)={int((jac[0] != 0).sum())}")


ROLLING_CASES = {
    # name: dict(nx, nu, hidden, H, kind, window, forward, box, B, p_dim, tvp_dim)
    "roll2_discret": dict(nx=2, nu=1, hidden=[32, 32], H=8, kind=orc.DISCRET, window=2, forward=True, box=None, B=3),
    "roll3_unity_rev": dict(nx=2, nu=1, hidden=[24], H=7, kind=orc.UNITY, window=3, forward=False, box=(-2.0, 2.0), B=2),
    "roll3_discret_rev": dict(nx=2, nu=1, hidden=[32, 32], H=6, kind=orc.DISCRET, window=3, forward=False, box=None, B=2),
    "roll4_wide": dict(nx=3, nu=2, hidden=[40, 40], H=6, kind=orc.DISCRET, window=4, forward=True, box=None, B=2),
    "roll2_tvp_p": dict(nx=2, nu=1, hidden=[32, 32], H=6, kind=orc.DISCRET, window=2, forward=True, box=None, B=2,
                        p_dim=1, tvp_dim=1),
    "roll4_short": dict(nx=2, nu=1, hidden=[16, 16], H=2, kind=orc.DISCRET, window=4, forward=True, box=None, B=2),
}


def build_rolling_case(ref, plugins, name, c):
    """Rolling-window models under the reference's own DiscretIntegrator / UnityIntegrator / IpoptProblem."""
    _, QuadObjective, BoxStateRows, NumpyRollingMLPModel = plugins
    nx, nu, H, kind, w, fwd, box, B = c["nx"], c["nu"], c["H"], c["kind"], c["window"], c["forward"], c["box"], c["B"]
    p_dim, tvp_dim = c.get("p_dim", 0), c.get("tvp_dim", 0)
    net = orc.MLP.random(w * (nx + nu) + w * tvp_dim + p_dim, c["hidden"], nx, seed=0)
    check_network_derivatives_by_ad(net, w * (nx + nu) + w * tvp_dim + p_dim, with_hess=max(c["hidden"]) <= 64)
    rng = np.random.default_rng(7)
    xref = rng.normal(size=(H, nx)) * 0.3
    uref = rng.normal(size=(H, nu)) * 0.3
    cu = rng.normal(size=(H, nu)) * 0.1
    Q = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx))
    Rm = 0.1 * np.eye(nu)
    pvec = rng.normal(size=p_dim) if p_dim else None
    tvp = rng.normal(size=(H, tvp_dim)) if tvp_dim else None
    prev_tvp = rng.normal(size=(w - 1, tvp_dim)) if tvp_dim else None
    hist_x = rng.normal(size=(B, w - 1, nx))
    hist_u = rng.uniform(-1.0, 1.0, size=(B, w - 1, nu))
    model = NumpyRollingMLPModel(net, nx, nu, p_dim, tvp_dim, rolling_window=w, forward_rolling=fwd)
    integ = ref.discret.DiscretIntegrator(model, H) if kind == orc.DISCRET else ref.unity.UnityIntegrator(model, H)
    ctrs = [BoxStateRows(np.full(nx, box[0]), np.full(nx, box[1]), nx, nu)] if box is not None else []
    # objective values come from the closed-form family; the network argument of Problem is unused by them
    obj = QuadObjective(orc.Problem(orc.MLP.random(nx + nu, [4], nx), H, nx, nu, Q=Q, R=Rm, xref=xref, uref=uref, cu=cu))
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    m = H * nx * (2 if box is not None else 1)
    lam = np.random.default_rng(3).normal(size=(B, m))
    sigma = np.array([1.0, 0.5, 2.0, 0.0])[:B]

    out = {"nx": nx, "nu": nu, "H": H, "kind": kind, "DT": 1.0, "hidden": np.array(c["hidden"]), "window": w,
           "forward_rolling": int(fwd), "hist_x": hist_x, "hist_u": hist_u,
           "Z": Z, "X0": X0, "Q": Q, "R": Rm, "xref": xref, "uref": uref, "cu": cu,
           "lam": lam, "sigma": sigma, "has_box": int(box is not None), "p_dim": p_dim, "tvp_dim": tvp_dim}
    if p_dim:
        out["p"] = pvec
    if tvp_dim:
        out["tvp"], out["prev_tvp"] = tvp, prev_tvp
    if box is not None:
        out["box_lo"], out["box_hi"] = np.full(nx, box[0]), np.full(nx, box[1])
    for i, (wt, b) in enumerate(zip(net.W, net.b)):
        out[f"W{i}"], out[f"b{i}"] = wt, b

    f, grad, g, jac, hvals, hdense = [], [], [], [], [], []
    for b in range(B):
        model.set_prev_data(hist_x[b], hist_u[b], tvp_prev=prev_tvp)
        pb = ref.ipopt.IpoptProblem(X0[b], obj, ctrs, integ, p=pvec, tvp=tvp)
        f.append(pb.objective(Z[b]))
        grad.append(pb.gradient(Z[b]))
        g.append(pb.constraints(Z[b]))
        jac.append(pb.jacobian(Z[b]))
        states, u = Z[b][:H * nx].reshape(H, nx), Z[b][H * nx:].reshape(H, nu)
        np.random.seed(11)
        rows, cols = pb.hessianstructure()
        out["h_rows"], out["h_cols"] = rows, cols
        hvals.append(pb.hessian(Z[b], lam[b], sigma[b]))
        ih = integ.hessian(states, u, X0[b], p=pvec, tvp=tvp)
        hdense.append(sigma[b] * obj.hessian(states, u) + np.einsum("i,ipq->pq", lam[b][:H * nx], ih))
    out.update(f=np.array(f), grad=np.array(grad), g=np.array(g), jac=np.array(jac),
               hvals=np.array(hvals), hdense=np.array(hdense))
    out["cl"], out["cu_bound"] = pb.get_constraint_lower_bounds(), pb.get_constraint_upper_bounds()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: w={w} n={H * (nx + nu)} m={m} B={B} nnz(jac)={int((jac[0] != 0).sum())}")


def build_misc(ref, plugins):
    """Bounds vectors, warm-start shift, and one full NMPC.next trajectory through reference SLSQP."""
    NumpyMLPModel, QuadObjective = plugins[:2]
    out = {}
    dc = ref.constraints.DomainConstraint(states_constraint=[[-np.inf, 1.0], [-2.0, np.inf]],
                                          control_constraint=[[-1.0, 0.2]])
    out["dom_lb"], out["dom_ub"] = np.array(dc.get_lower_bounds(5)), np.array(dc.get_upper_bounds(5))

    nx, nu, H = 2, 1, 10
    net = orc.MLP.random(nx + nu, [30, 30], nx, seed=0)
    # damp the net so the closed loop is well behaved: x+ = x + 0.2 f(x,u)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET, 1.0, Q=np.eye(nx), R=0.1 * np.eye(nu))
    model = NumpyMLPModel(net, nx, nu)
    integ = ref.discret.DiscretIntegrator(model, H)
    obj = QuadObjective(prob)
    dom = ref.constraints.DomainConstraint(states_constraint=[[-5.0, 5.0]] * nx, control_constraint=[[-1.0, 1.0]] * nu)
    opt = ref.slsqp.Slsqp(max_iteration=200, tolerance=1e-10, verbose=0, init_with_last_result=True)
    mpc = ref.controller.NMPC(integ, obj, [dom], H, 1.0, optimizer=opt)
    x0 = np.array([0.7, -0.4])
    # record the initial guess the reference hands to scipy (cold start slsqp.py:163, warm start slsqp.py:155-161)
    inits = []
    real_minimize = ref.slsqp.minimize

    def recording_minimize(fun, x_init, *a, **k):
        inits.append(np.array(x_init, dtype=np.float64))
        return real_minimize(fun, x_init, *a, **k)

    ref.slsqp.minimize = recording_minimize
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        states, u = mpc.next(x0)
        first = opt.prev_result.copy()
        x1 = states[0].copy()
        states2, u2 = mpc.next(x1)   # warm-start branch
    ref.slsqp.minimize = real_minimize
    out.update(nmpc_x0=x0, nmpc_states=states, nmpc_u=u, nmpc_prev=first, nmpc_x1=x1,
               nmpc_states2=states2, nmpc_u2=u2, nmpc_H=H, cold_init=inits[0], warm_from_first=inits[1])
    for i, (w, b) in enumerate(zip(net.W, net.b)):
        out[f"W{i}"], out[f"b{i}"] = w, b
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
    print("misc: NMPC.next ok, max|g| at solution =",
          float(np.abs(integ.forward(states, u, x0)).max()))


def main():
    ref = import_reference()
    plugins = make_plugins(ref)
    only = sys.argv[1:]
    for name, spec in CASES.items():
        if not only or name in only:
            build_case(ref, plugins, name, spec)
    for name, spec in ROLLING_CASES.items():
        if not only or name in only or "rolling" in only:
            build_rolling_case(ref, plugins, name, spec)
    if not only or "misc" in only:
        build_misc(ref, plugins)


if __name__ == "__main__":
    main()
