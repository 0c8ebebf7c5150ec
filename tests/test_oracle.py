"""CPU tests: the oracle against the reference-generated golden vectors, and the network
derivative (the piece the reference delegates to TF/JAX autodiff) against torch AD and FD."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc
from helpers import ACT_MIXED_NAMES, ACT_UNIFORM_NAMES, CASE_NAMES, ROLLING_NAMES, WIDE_DEEP_NAMES, ZBASED_NAMES, load_case, oracle_problem

TOL = dict(rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", CASE_NAMES + ACT_UNIFORM_NAMES + ACT_MIXED_NAMES + WIDE_DEEP_NAMES + ZBASED_NAMES)
def test_oracle_matches_reference_golden(name):
    d, W, b = load_case(name)
    prob = oracle_problem(d, W, b)
    B = d["Z"].shape[0]
    for i in range(B):
        z, x0 = d["Z"][i], d["X0"][i]
        np.testing.assert_allclose(prob.objective(z), d["f"][i], **TOL)
        np.testing.assert_allclose(prob.gradient(z), d["grad"][i], **TOL)
        np.testing.assert_allclose(prob.constraints(z, x0), d["g"][i], **TOL)
        np.testing.assert_allclose(prob.jacobian(z, x0), d["jac"][i], **TOL)
        # integrator-only outputs (no glue)
        nxh = prob.H * prob.nx
        np.testing.assert_allclose(prob.constraints(z, x0)[:nxh], d["g_int"][i], **TOL)
        np.testing.assert_allclose(prob.jacobian(z, x0)[:nxh], d["jac_int"][i], **TOL)
    cl, cu = prob.constraint_bounds()
    np.testing.assert_array_equal(cl, d["cl"])
    np.testing.assert_array_equal(cu, d["cu_bound"])
    # structural zeros are exact zeros in both
    assert np.array_equal(prob.jacobian(d["Z"][0], d["X0"][0]) != 0, d["jac"][0] != 0)


@pytest.mark.parametrize("name", [n for n in CASE_NAMES if n not in ("c3_rk4", "c3_discret", "odd_dims")] +
                         [n for n in ACT_UNIFORM_NAMES if n.endswith("_c2")] + ACT_MIXED_NAMES + WIDE_DEEP_NAMES + ZBASED_NAMES)
def test_oracle_hessian_matches_reference_golden(name):
    d, W, b = load_case(name)
    prob = oracle_problem(d, W, b)
    rows, cols = prob.hessian_structure()
    ours = set(zip(rows.tolist(), cols.tolist()))
    ref_pat = set(zip(d["h_rows"].tolist(), d["h_cols"].tolist()))
    assert ref_pat <= ours, "reference's sampled pattern must be inside the structural pattern"
    for i in range(d["Z"].shape[0]):
        Hm = prob.lagrangian_hessian(d["Z"][i], d["X0"][i], d["lam"][i], float(d["sigma"][i]))
        np.testing.assert_allclose(Hm, d["hdense"][i], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Hm[d["h_rows"], d["h_cols"]], d["hvals"][i], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Hm, Hm.T, rtol=1e-11, atol=1e-13)
        # nothing outside the structural pattern
        mask = np.zeros_like(Hm, dtype=bool)
        mask[rows, cols] = True
        assert np.all(np.tril(Hm)[~mask] == 0.0)


@pytest.mark.parametrize("name", ROLLING_NAMES)
def test_oracle_rolling_window_matches_reference_golden(name):
    """Rolling-window models under the reference's DiscretIntegrator / UnityIntegrator / IpoptProblem."""
    d, W, b = load_case(name)
    for i in range(d["Z"].shape[0]):
        prob = oracle_problem(d, W, b, i)
        z, x0 = d["Z"][i], d["X0"][i]
        np.testing.assert_allclose(prob.objective(z), d["f"][i], **TOL)
        np.testing.assert_allclose(prob.gradient(z), d["grad"][i], **TOL)
        np.testing.assert_allclose(prob.constraints(z, x0), d["g"][i], **TOL)
        np.testing.assert_allclose(prob.jacobian(z, x0), d["jac"][i], **TOL)
        assert np.array_equal(prob.jacobian(z, x0) != 0, d["jac"][i] != 0)
        Hm = prob.lagrangian_hessian(z, x0, d["lam"][i], float(d["sigma"][i]))
        np.testing.assert_allclose(Hm, d["hdense"][i], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Hm[d["h_rows"], d["h_cols"]], d["hvals"][i], rtol=1e-11, atol=1e-12)
        rows, cols = prob.hessian_structure()
        assert set(zip(d["h_rows"].tolist(), d["h_cols"].tolist())) <= set(zip(rows.tolist(), cols.tolist()))
        mask = np.zeros_like(Hm, dtype=bool)
        mask[rows, cols] = True
        assert np.all(np.tril(Hm)[~mask] == 0.0)
    cl, cu = prob.constraint_bounds()
    np.testing.assert_array_equal(cl, d["cl"])
    np.testing.assert_array_equal(cu, d["cu_bound"])


def test_rolling_window_derivatives_vs_finite_differences():
    d, W, b = load_case("roll3_discret_rev")
    prob = oracle_problem(d, W, b, 1)
    z, x0 = d["Z"][1], d["X0"][1]
    J = prob.jacobian(z, x0)
    lam = d["lam"][1]
    Hm = prob.lagrangian_hessian(z, x0, lam, 0.0)
    eps = 1e-6
    Jfd, Hfd = np.zeros_like(J), np.zeros_like(Hm)
    for j in range(prob.n):
        e = np.zeros(prob.n); e[j] = eps
        Jfd[:, j] = (prob.constraints(z + e, x0) - prob.constraints(z - e, x0)) / (2 * eps)
        Hfd[:, j] = (lam @ prob.jacobian(z + e, x0) - lam @ prob.jacobian(z - e, x0)) / (2 * eps)
    assert np.abs(J - Jfd).max() < 1e-8
    assert np.abs(Hm - Hfd).max() < 1e-7


def test_slsqp_glue_rows():
    d, W, b = load_case("c2_discret")
    prob = oracle_problem(d, W, b)
    np.testing.assert_allclose(prob.constraints(d["Z"][0], d["X0"][0]), d["slsqp_eq"], **TOL)
    np.testing.assert_allclose(prob.jacobian(d["Z"][0], d["X0"][0]), d["slsqp_eq_jac"], **TOL)


def test_bounds_and_warm_start_against_reference():
    m = dict(np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "misc.npz")))
    lb, ub = orc.domain_bounds([[-np.inf, 1.0], [-2.0, np.inf]], [[-1.0, 0.2]], 5)
    np.testing.assert_array_equal(lb, m["dom_lb"])
    np.testing.assert_array_equal(ub, m["dom_ub"])
    H = int(m["nmpc_H"])
    np.testing.assert_array_equal(orc.cold_start(m["nmpc_x0"], H, 1), m["cold_init"])
    np.testing.assert_array_equal(orc.warm_start_shift(m["nmpc_prev"], H, 2, 1), m["warm_from_first"])


# ---- the network derivative: independent AD (torch.func, fp64) ----
def _torch_act(spec):
    """torch's OWN activation for an oracle activation spec ("name" | "name:alpha")"""
    name, par = orc.act_split(spec)
    F = torch.nn.functional
    return {"linear": lambda z: z, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, "softplus": F.softplus,
            "elu": lambda z: F.elu(z, alpha=par), "leaky_relu": lambda z: F.leaky_relu(z, negative_slope=par), "selu": F.selu,
            "swish": F.silu, "gelu": F.gelu, "softsign": F.softsign, "mish": F.mish, "exponential": torch.exp, "relu6": F.relu6}[name]


def _torch_net(W, b, act=None):
    Wt = [torch.tensor(w, dtype=torch.float64) for w in W]
    bt = [torch.tensor(x, dtype=torch.float64) for x in b]
    act = ["tanh"] * (len(W) - 1) + ["linear"] if act is None else act

    def f(xi):
        a = xi
        for w, bb, name in zip(Wt, bt, act):
            a = _torch_act(name)(a @ w + bb)
        return a
    return f


@pytest.mark.parametrize("acts", ["relu", "sigmoid", "softplus", "elu", ["relu", "softplus", "sigmoid"],
                                  ["elu", "sigmoid", "tanh"], ["linear", "tanh", "softplus"], ["sigmoid", "elu", "elu"],
                                  "selu", "leaky_relu", "elu:0.5", ["leaky_relu:0.05", "selu", "elu:1.7"],
                                  ["selu", "leaky_relu:0.3", "selu"], "swish", "gelu", ["swish", "gelu", "tanh"],
                                  ["gelu", "relu", "swish"], "softsign", "mish", "exponential", "relu6",
                                  ["mish", "softsign", "relu6"], ["exponential", "relu6", "mish"]])
def test_activation_family_derivatives_vs_torch_ad(acts):
    """Every activation of the device family, uniform on the hidden layers and mixed per layer with a non-linear output
    layer: the oracle's first and second derivatives -- written from the layer OUTPUT a = s(z), as the kernels do (from the
    pre-activation for the non-monotone swish / gelu) -- against
    torch's own activations under torch.func autodiff (fp64).  This is the guard the fixtures of these activations rest
    on (make_golden.check_network_derivatives_by_ad runs the same comparison while they are made)."""
    nin, hidden, nout = 5, [24, 17], 3
    net = orc.MLP.random(nin, hidden, nout, seed=4, activations=acts)
    xi = np.random.default_rng(5).normal(size=(6, nin)) * 1.5
    f, J, S = net.forward_jac_hess(xi)
    f2, J2 = net.forward_jac(xi)
    np.testing.assert_allclose(f, f2, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(J, J2, rtol=1e-12, atol=1e-13)
    tf = _torch_net(net.W, net.b, net.act)
    for r in range(xi.shape[0]):
        x = torch.tensor(xi[r], dtype=torch.float64)
        np.testing.assert_allclose(tf(x).numpy(), f[r], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(torch.func.jacrev(tf)(x).numpy(), J[r], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(torch.func.hessian(tf)(x).numpy(), S[r], rtol=1e-10, atol=1e-12)


def test_activation_functions_at_their_edges():
    """Values the formulas must survive: large |z| (no overflow / NaN), the kinks, NaN propagation."""
    z = np.array([-800.0, -40.0, -1e-300, 0.0, 1e-300, 40.0, 800.0])
    for name in orc.ACTIVATIONS:
        if name == "exponential":
            z = np.minimum(z, 40.0)          # (e^800 IS infinite: the function's own range, not a formula to survive)
        a = orc.act_f(name, z)
        assert np.all(np.isfinite(a)), name
        assert np.all(np.isfinite(orc.act_s1(name, z, a))) and np.all(np.isfinite(orc.act_s2(name, z, a))), name
        if name not in orc.ZBASED:
            assert np.all(np.isfinite(orc.act_d1(name, a))) and np.all(np.isfinite(orc.act_r2(name, a))), name
        assert np.isnan(orc.act_f(name, np.array([np.nan]))[0]), name
    assert orc.act_d1("relu", orc.act_f("relu", np.array([0.0])))[0] == 0.0          # TensorFlow's convention at the kink
    assert orc.act_d1("elu", orc.act_f("elu", np.array([0.0])))[0] == 1.0
    assert orc.act_d1("elu:0.5", orc.act_f("elu:0.5", np.array([0.0])))[0] == 0.5
    assert orc.act_d1("leaky_relu:0.1", orc.act_f("leaky_relu:0.1", np.array([0.0])))[0] == 0.1    # tf.nn.leaky_relu's gradient at 0
    for bad in ("elu:0", "elu:-1", "leaky_relu:-0.1", "tanh:2", "hard_sigmoid", "gelu:1"):
        with pytest.raises(ValueError):
            orc.act_split(bad)
    np.testing.assert_allclose(orc.act_d1("softplus", orc.act_f("softplus", z)), 1.0 / (1.0 + np.exp(-np.clip(z, -700, 700))),
                               rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("dims", [(3, [64, 64], 2), (9, [128, 128, 128], 6), (5, [48, 32], 3), (3, [16], 2)])
def test_mlp_derivatives_vs_torch_ad(dims):
    nin, hidden, nout = dims
    net = orc.MLP.random(nin, hidden, nout, seed=4)
    xi = np.random.default_rng(5).normal(size=(6, nin))
    f, J, S = net.forward_jac_hess(xi)
    f2, J2 = net.forward_jac(xi)
    np.testing.assert_allclose(f, f2, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(J, J2, rtol=1e-12, atol=1e-13)
    tf = _torch_net(net.W, net.b)
    for r in range(xi.shape[0]):
        x = torch.tensor(xi[r], dtype=torch.float64)
        np.testing.assert_allclose(tf(x).numpy(), f[r], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(torch.func.jacrev(tf)(x).numpy(), J[r], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(torch.func.hessian(tf)(x).numpy(), S[r], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("kind,DT", [(orc.DISCRET, 1.0), (orc.UNITY, 1.0), (orc.RK4, 0.1), (orc.RK4, 0.5)])
def test_step_derivatives_vs_torch_ad(kind, DT):
    nx, nu = 3, 2
    net = orc.MLP.random(nx + nu, [24, 24], nx, seed=2)
    rng = np.random.default_rng(9)
    xp, u = rng.normal(size=(4, nx)), rng.normal(size=(4, nu))
    phi, dphi, d2phi = orc.step_rows(net, kind, DT, xp, u, want_hess=True)
    tf = _torch_net(net.W, net.b)

    def step(xi):
        x, uu = xi[:nx], xi[nx:]
        f = lambda xs: tf(torch.cat([xs, uu]))
        if kind == orc.DISCRET:
            return x + f(x)
        if kind == orc.UNITY:
            return f(x)
        k1 = f(x); k2 = f(x + 0.5 * DT * k1); k3 = f(x + 0.5 * DT * k2); k4 = f(x + DT * k3)
        return x + DT / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    for r in range(4):
        xi = torch.tensor(np.concatenate([xp[r], u[r]]), dtype=torch.float64)
        np.testing.assert_allclose(step(xi).numpy(), phi[r], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(torch.func.jacrev(step)(xi).numpy(), dphi[r], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(torch.func.hessian(step)(xi).numpy(), d2phi[r], rtol=1e-9, atol=1e-11)


def test_jacobian_and_gradient_vs_finite_differences():
    d, W, b = load_case("c2_rk4")
    prob = oracle_problem(d, W, b)
    z, x0 = d["Z"][0], d["X0"][0]
    J = prob.jacobian(z, x0)
    gr = prob.gradient(z)
    eps = 1e-6
    Jfd = np.zeros_like(J)
    gfd = np.zeros_like(gr)
    for j in range(prob.n):
        e = np.zeros(prob.n); e[j] = eps
        Jfd[:, j] = (prob.constraints(z + e, x0) - prob.constraints(z - e, x0)) / (2 * eps)
        gfd[j] = (prob.objective(z + e) - prob.objective(z - e)) / (2 * eps)
    assert np.abs(J - Jfd).max() < 1e-8
    assert np.abs(gr - gfd).max() < 1e-7


@pytest.mark.parametrize("name", ["c2_discret", "c2_unity", "c3_rk4", "c5_box", "odd_dims", "h1", "act_relu_c2",
                                  "act_sigmoid_c3", "act_softplus_c2", "act_elu_c3", "act_mixed_box", "act_mixed_rk4",
                                  "act_linear_hidden", "act_param_box", "act_selu_rk4", "act_swish_gelu_box", "act_gelu_rk4"])
def test_c_oracle_matches_numpy_oracle(name):
    from oracle.c_oracle import COracle
    d, W, b = load_case(name)
    prob = oracle_problem(d, W, b)
    Z, X0 = orc.synthetic_inputs(5, prob.H, prob.nx, prob.nu, seed=21)
    f, grad, g, jac = prob.eval_batch(Z, X0)
    cf, cgrad, cg, cjac = COracle(prob).eval(Z, X0)
    np.testing.assert_allclose(cf, f, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cgrad, grad, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cg, g, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cjac, jac, rtol=1e-11, atol=1e-12)


def test_gauss_newton_hessian_is_the_first_order_part_of_the_exact_one():
    """oracle.gauss_newton_hessian: (a) its per-step blocks are T^T diag(w) T with T checked by central finite
    differences of the step map; (b) for a network without hidden layer (zero second derivative) the exact Lagrangian
    Hessian has no constraint curvature at all, and with w = 0 both reduce to sigma * d2f; (c) PSD for w >= 0."""
    H, nx, nu = 5, 2, 1
    net = orc.MLP.random(nx + nu, [12, 9], nx, seed=4)
    prob = orc.Problem(net, H, nx, nu, orc.RK4, 0.2, Q=np.diag([1.0, 2.0]), R=np.array([[0.3]]))
    Z, X0 = orc.synthetic_inputs(1, H, nx, nu, seed=2)
    z, x0 = Z[0], X0[0]
    w = np.random.default_rng(5).uniform(0.2, 1.5, size=H * nx)
    G = prob.gauss_newton_hessian(z, x0, w, 0.8)
    eps = 1e-6
    Jfd = np.zeros((H * nx, prob.n))
    for j in range(prob.n):
        e = np.zeros(prob.n); e[j] = eps
        Jfd[:, j] = (prob.constraints(z + e, x0) - prob.constraints(z - e, x0)) / (2 * eps)
    ref = 0.8 * prob.objective_hessian()
    for t in range(H):
        cols = prob.tile_columns(t)
        keep = cols[cols >= 0]
        T = Jfd[t * nx:(t + 1) * nx][:, keep].copy()
        # the defect's own -x_t column is not part of the tile; tile columns are x_{t-1} and u_t only
        ref[np.ix_(keep, keep)] += T.T @ np.diag(w[t * nx:(t + 1) * nx]) @ T
    assert np.abs(G - ref).max() < 1e-7
    assert np.array_equal(G, G.T) or np.abs(G - G.T).max() < 1e-15
    assert np.linalg.eigvalsh(0.5 * (G + G.T)).min() > -1e-12
    lin = orc.MLP([np.random.default_rng(1).normal(size=(nx + nu, nx))], [np.zeros(nx)])
    pl = orc.Problem(lin, H, nx, nu, orc.DISCRET)
    lam = np.random.default_rng(2).normal(size=pl.m)
    np.testing.assert_allclose(pl.lagrangian_hessian(z, x0, lam, 0.8), pl.gauss_newton_hessian(z, x0, np.zeros(H * nx), 0.8),
                               atol=1e-14)
    r, c = prob.hessian_structure()
    mask = np.zeros((prob.n, prob.n), dtype=bool); mask[r, c] = True; mask |= mask.T
    assert np.all(G[~mask] == 0.0)       # same pattern as the exact callback
