"""GPU parity for rolling-window models (KerasTFModelRollingInput model/tensorflow.py:132-340,
DiffDiscretJaxModelRollingWindow model/jax.py:93-259): the HIP path through the C ABI against golden vectors
produced by the reference's DiscretIntegrator / UnityIntegrator / IpoptProblem, and against the oracle on seeded
batches with a different history per problem.  fp64 1e-12; fp32 1e-4 relative."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc
from helpers import ROLLING_NAMES, case_extra, load_case, oracle_problem

pytestmark = pytest.mark.gpu

F64 = dict(rtol=1e-12, atol=1e-12)
KIND_NAME = {0: "discret", 1: "unity", 2: "rk4"}
ALL = ("f", "grad", "g", "jac_dense", "jac_tiles", "jac_sparse")


def _kernels(d):
    tw = int(d["window"]) * (int(d["nx"]) + int(d["nu"]))
    ex = case_extra(d)
    fits = tw + (0 if ex is None else ex.shape[1]) <= 32 and tw <= 32     # matrix-core kernels: up to 32 network inputs
    # (the layer-at-a-time path takes windows of up to 32 decision inputs, any number of extra inputs)
    return (["valu", "mfma", "mfma_tile"] if fits else ["valu"]) + (["layered"] if tw <= 32 else [])


def _engine(d, W, b, dtype, kernel):
    from pyneuralempc_amd import CallbackEngine
    ex = case_extra(d)
    B = d["Z"].shape[0]
    eng = CallbackEngine(W, b, int(d["H"]), int(d["nx"]), int(d["nu"]), integrator=KIND_NAME[int(d["kind"])],
                         dtype=dtype, device="cuda:0", max_batch=B, kernel=kernel,
                         n_extra=0 if ex is None else ex.shape[1], rolling_window=int(d["window"]),
                         forward_rolling=bool(int(d["forward_rolling"])))
    if ex is not None:
        eng.bind_extra(eng.to_device(np.broadcast_to(ex[None], (B,) + ex.shape).copy()))
    eng.bind_history(eng.to_device(d["hist_x"]), eng.to_device(d["hist_u"]))
    eng.set_objective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"])
    if int(d["has_box"]):
        eng.set_box_rows(d["box_lo"], d["box_hi"])
    return eng


@pytest.mark.parametrize("name", ROLLING_NAMES)
def test_rolling_golden_fp64(name):
    d, W, b = load_case(name)
    for kernel in _kernels(d):
        eng = _engine(d, W, b, torch.float64, kernel)
        assert eng.kernel_variant == kernel
        res = eng.eval_numpy(d["Z"], d["X0"], want=ALL)
        np.testing.assert_allclose(res["f"], d["f"], **F64)
        np.testing.assert_allclose(res["grad"], d["grad"], **F64)
        np.testing.assert_allclose(res["g"], d["g"], **F64)
        np.testing.assert_allclose(res["jac_dense"], d["jac"], **F64)
        assert np.array_equal(res["jac_dense"] != 0, d["jac"] != 0)
        rows, cols = eng.jac_structure()
        assert np.array_equal(res["jac_sparse"], res["jac_dense"][:, rows, cols])
        assert np.all(np.diff(rows.astype(np.int64) * eng.n + cols) > 0)     # row-major sorted, no duplicates
        for i in range(d["Z"].shape[0]):
            _, _, dphi, _ = oracle_problem(d, W, b, i).tiles(d["Z"][i], d["X0"][i])
            np.testing.assert_allclose(res["jac_tiles"][i], dphi, **F64)
        cl, cu = eng.constraint_bounds()
        np.testing.assert_array_equal(cl, d["cl"])
        np.testing.assert_array_equal(cu, d["cu_bound"])
        # both matrix-core kernels serve rolling windows: "mfma" resolves to the cooperative one when it fits
        expect = {"valu": "rows_valu_kernel", "mfma_tile": "rows_mfma_kernel", "layered": "layered_gemm_kernel"}.get(kernel)
        if expect is not None:
            assert eng.last_row_kernel == expect
        else:
            assert eng.last_row_kernel in ("rows_coop_kernel", "rows_mfma_kernel")


@pytest.mark.parametrize("name", ROLLING_NAMES)
def test_rolling_golden_hessian_fp64(name):
    d, W, b = load_case(name)
    for kernel in _kernels(d):
        eng = _engine(d, W, b, torch.float64, kernel)
        out = eng.hess(eng.to_device(d["Z"]), eng.to_device(d["X0"]), eng.to_device(d["lam"]),
                       eng.to_device(d["sigma"]), want=("hvals", "hdense", "hblocks"))
        hd, hv = out["hdense"].cpu().numpy(), out["hvals"].cpu().numpy()
        np.testing.assert_allclose(hd, d["hdense"], rtol=1e-11, atol=1e-12)
        rows, cols = eng.hess_structure()
        assert np.array_equal(hv, hd[:, rows, cols])
        np.testing.assert_allclose(hd[:, d["h_rows"], d["h_cols"]], d["hvals"], rtol=1e-11, atol=1e-12)
        orows, ocols = oracle_problem(d, W, b).hessian_structure()
        assert np.array_equal(rows, orows) and np.array_equal(cols, ocols)
        assert np.array_equal(hd, np.transpose(hd, (0, 2, 1)))
        mask = np.zeros(hd.shape[1:], dtype=bool)
        mask[rows, cols] = True
        assert np.all(np.tril(hd)[:, ~mask] == 0.0)


@pytest.mark.parametrize("name", ["roll2_discret", "roll3_unity_rev", "roll4_wide"])
def test_rolling_golden_fp32(name):
    d, W, b = load_case(name)
    for kernel in _kernels(d):
        eng = _engine(d, W, b, torch.float32, kernel)
        res = eng.eval_numpy(d["Z"], d["X0"], want=ALL)
        for k, ref in (("f", d["f"]), ("grad", d["grad"]), ("g", d["g"]), ("jac_dense", d["jac"])):
            scale = max(1.0, np.abs(ref).max())
            assert np.abs(res[k] - ref).max() / scale < 1e-4, f"{name}/{kernel}/{k}"
        assert np.array_equal(res["jac_dense"] != 0, d["jac"] != 0)


@pytest.mark.parametrize("cfg", [(2, 1, [64, 64], 20, 2, True, 70), (2, 1, [64, 64], 20, 4, False, 37),
                                 (3, 1, [32, 32, 32], 9, 3, True, 19), (4, 2, [48], 5, 5, True, 6)])
def test_rolling_seeded_batches_against_oracle(cfg):
    """Bigger batches (ragged last tile, per-problem histories) against the oracle; C2 dims with a window."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, w, fwd, B = cfg
    tw = w * (nx + nu)
    net = orc.MLP.random(tw, hidden, nx, seed=3)
    rng = np.random.default_rng(11)
    hx, hu = rng.normal(size=(B, w - 1, nx)), rng.uniform(-1, 1, size=(B, w - 1, nu))
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    lamh = rng.normal(size=(B, H * nx))
    sigh = rng.uniform(0.0, 2.0, size=B)
    kernels = ["valu", "mfma"] if tw <= 32 else ["valu"]
    for kernel in kernels:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel,
                             rolling_window=w, forward_rolling=fwd)
        eng.bind_history(eng.to_device(hx), eng.to_device(hu))
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        out = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lamh), eng.to_device(sigh),
                       want=("hdense",))
        hd = out["hdense"].cpu().numpy()
        for i in list(range(0, B, max(1, B // 5))) + [B - 1]:
            prob = orc.Problem(net, H, nx, nu, orc.DISCRET, window=w, forward_rolling=fwd, hist_x=hx[i], hist_u=hu[i])
            np.testing.assert_allclose(res["g"][i], prob.constraints(Zh[i], X0h[i]), **F64)
            np.testing.assert_allclose(res["jac_dense"][i], prob.jacobian(Zh[i], X0h[i]), **F64)
            np.testing.assert_allclose(hd[i], prob.lagrangian_hessian(Zh[i], X0h[i], lamh[i], sigh[i]),
                                       rtol=1e-10, atol=1e-11)


def test_rolling_window_of_one_is_the_plain_model():
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 6, 5
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=1)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=2)
    a = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B)
    b = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B, rolling_window=1, forward_rolling=False)
    ra, rb = a.eval_numpy(Zh, X0h), b.eval_numpy(Zh, X0h)
    for k in ra:
        assert np.array_equal(ra[k], rb[k])


def test_rolling_errors():
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd._lib import NempcError
    nx, nu, H = 2, 1, 4
    net = orc.MLP.random(2 * (nx + nu), [16], nx, seed=1)
    with pytest.raises(NempcError, match="DISCRET or UNITY"):
        CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, device="cuda:0", rolling_window=2)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=2, rolling_window=2)
    Z, X0 = (eng.to_device(a) for a in orc.synthetic_inputs(2, H, nx, nu))
    with pytest.raises(ValueError, match="set_prev_data"):
        eng.eval(Z, X0)
    with pytest.raises(ValueError, match="hist_x"):
        eng.bind_history(torch.zeros(2, 2, nx, dtype=torch.float64, device="cuda:0"),
                         torch.zeros(2, 1, nu, dtype=torch.float64, device="cuda:0"))
    eng.bind_history(torch.zeros(2, 1, nx, dtype=torch.float64, device="cuda:0"),
                     torch.zeros(2, 1, nu, dtype=torch.float64, device="cuda:0"))
    eng.eval(Z, X0)
    eng.bind_history(torch.zeros(1, 1, nx, dtype=torch.float64, device="cuda:0"),
                     torch.zeros(1, 1, nu, dtype=torch.float64, device="cuda:0"))
    with pytest.raises(ValueError, match="set_prev_data"):       # the history must cover the batch that is solved
        eng.solve(X0)


@pytest.mark.parametrize("cfg", [(2, 1, [32, 32], 12, 2, True, "valu", "none", 24), (2, 1, [32, 32], 12, 3, False, "mfma", "controls", 40),
                                 (3, 2, [24], 6, 2, True, "mfma", "box", 9), (1, 1, [16, 16], 8, 4, True, "valu", "controls", 7)])
def test_batched_solve_of_rolling_window_models(cfg):
    """nempc_solve on rolling-window models: the window is made the state and the same Riccati SQP runs on it.  Checked on
    the CALLER's problem with the oracle: feasibility, first-order optimality, and SciPy SLSQP on the oracle's callbacks."""
    from scipy.optimize import Bounds, minimize
    import warnings
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, w, fwd, kernel, bounds, B = cfg
    net = orc.MLP.random(w * (nx + nu), hidden, nx, seed=4)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    rng = np.random.default_rng(21)
    hx, hu = rng.uniform(-1, 1, size=(B, w - 1, nx)), rng.uniform(-0.3, 0.3, size=(B, w - 1, nu))
    X0 = rng.uniform(-1.0, 1.0, size=(B, nx))
    Q, R = np.eye(nx), 0.1 * np.eye(nu)
    n = H * (nx + nu)
    lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    if bounds != "none":
        lb[H * nx:], ub[H * nx:] = -0.05, 0.05
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel,
                         rolling_window=w, forward_rolling=fwd)
    eng.set_objective(Q=Q, R=R)
    if bounds == "box":
        eng.set_box_rows(-3.0, 3.0)
        lb[:H * nx], ub[:H * nx] = -3.0, 3.0
    eng.bind_history(eng.to_device(hx), eng.to_device(hu))
    Z, status, iters, its = eng.solve(eng.to_device(X0), lb=lb if bounds != "none" else None, ub=ub if bounds != "none" else None,
                                      max_iter=400, return_iterations=True)
    Z, status = Z.cpu().numpy(), status.cpu().numpy()
    assert (status == 0).all(), f"{int((status != 0).sum())} of {B} problems did not converge in {iters} iterations"
    assert int(its.max()) <= iters
    same = 0
    for i in range(B):
        prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=Q, R=R, window=w, forward_rolling=fwd, hist_x=hx[i], hist_u=hu[i])
        assert np.abs(prob.constraints(Z[i], X0[i])[:H * nx]).max() < 1e-7
        assert (Z[i] >= lb - 1e-12).all() and (Z[i] <= ub + 1e-12).all()
        if i % max(1, B // 5):
            continue
        J, gr = prob.jacobian(Z[i], X0[i])[:H * nx], prob.gradient(Z[i])
        free = (Z[i] > lb + 1e-3) & (Z[i] < ub - 1e-3)
        lam = np.linalg.lstsq(J[:, free].T, -gr[free], rcond=None)[0]
        assert np.abs(gr[free] + J[:, free].T @ lam).max() < 1e-5 * max(1.0, np.abs(gr).max())
        rest = gr + J.T @ lam
        assert (rest[Z[i] <= lb + 1e-3] > -1e-3).all() and (rest[Z[i] >= ub - 1e-3] < 1e-3).all()
        zi = np.clip(orc.cold_start(X0[i], H, nu), np.where(np.isfinite(lb), lb + 1e-3, -np.inf),
                     np.where(np.isfinite(ub), ub - 1e-3, np.inf))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = minimize(prob.objective, zi, method="SLSQP", jac=prob.gradient, bounds=Bounds(lb, ub),
                           constraints=[{"type": "eq", "fun": lambda z: prob.constraints(z, X0[i])[:H * nx],
                                         "jac": lambda z: prob.jacobian(z, X0[i])[:H * nx]}],
                           options={"maxiter": 500, "ftol": 1e-12})
        f_gpu = prob.objective(Z[i])
        assert f_gpu <= ref.fun * (1 + 1e-6) + 1e-8, (f_gpu, ref.fun)
        if abs(f_gpu - ref.fun) <= 1e-5 * max(1.0, abs(ref.fun)):
            np.testing.assert_allclose(Z[i], ref.x, rtol=0, atol=2e-4)
            same += 1
    assert same >= 3, "most sampled problems should land in SLSQP's minimum"


# ---------------------------------------------------------------------------------------------------------------
# reference-shaped plug-in surface: Model.set_prev_data / forward / jacobian / hessian, Integrator, IpoptProblem, NMPC
# ---------------------------------------------------------------------------------------------------------------
def _rolling_model(d, W, b):
    import pyneuralempc_amd as nEMPC
    return nEMPC.model.MLPModelRollingInput(W, b, int(d["nx"]), int(d["nu"]), p_dim=int(d["p_dim"]),
                                            tvp_dim=int(d["tvp_dim"]), rolling_window=int(d["window"]),
                                            forward_rolling=bool(int(d["forward_rolling"])), device="cuda:0")


@pytest.mark.parametrize("name", ["roll2_discret", "roll3_unity_rev", "roll2_tvp_p", "roll4_wide"])
def test_rolling_plugins_like_reference(name):
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    d, W, b = load_case(name)
    H, nx, nu = int(d["H"]), int(d["nx"]), int(d["nu"])
    model = _rolling_model(d, W, b)
    integ = (nEMPC.integrator.discret.DiscretIntegrator(model, H) if int(d["kind"]) == orc.DISCRET
             else nEMPC.integrator.unity.UnityIntegrator(model, H))
    obj = nEMPC.objective.QuadraticObjective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"],
                                             device="cuda:0")
    ctrs = [nEMPC.constraints.BoxStateConstraint(d["box_lo"], d["box_hi"])] if int(d["has_box"]) else []
    p = d["p"] if int(d["p_dim"]) else None
    tvp = d["tvp"] if int(d["tvp_dim"]) else None
    states = d["Z"][0][:H * nx].reshape(H, nx)
    u = d["Z"][0][H * nx:].reshape(H, nu)
    with pytest.raises(AssertionError, match="set_prev_data"):
        integ.forward(states, u, d["X0"][0], p=p, tvp=tvp)
    nh = H * nx
    for i in range(d["Z"].shape[0]):
        model.set_prev_data(d["hist_x"][i], d["hist_u"][i], tvp_prev=d.get("prev_tvp"))
        z = d["Z"][i]
        states, u = z[:nh].reshape(H, nx), z[nh:].reshape(H, nu)
        np.testing.assert_allclose(integ.forward(states, u, d["X0"][i], p=p, tvp=tvp), d["g"][i][:nh], **F64)
        np.testing.assert_allclose(integ.jacobian(states, u, d["X0"][i], p=p, tvp=tvp), d["jac"][i][:nh], **F64)
        pb = IpoptProblem(d["X0"][i], obj, ctrs, integ, p=p, tvp=tvp)
        assert pb._fused is not None
        np.testing.assert_allclose(pb.objective(z), d["f"][i], **F64)
        np.testing.assert_allclose(pb.gradient(z), d["grad"][i], **F64)
        np.testing.assert_allclose(pb.constraints(z), d["g"][i], **F64)
        np.testing.assert_allclose(pb.jacobian(z), d["jac"][i], **F64)
        rows, cols = pb.hessianstructure()
        np.testing.assert_allclose(pb.hessian(z, d["lam"][i], float(d["sigma"][i])), d["hdense"][i][rows, cols],
                                   rtol=1e-11, atol=1e-12)
    # integrator Hessian tensor and its structure map
    ih = integ.hessian(states, u, d["X0"][i], p=p, tvp=tvp)
    lam = d["lam"][i][:nh]
    prob = oracle_problem(d, W, b, i)
    np.testing.assert_allclose(np.einsum("i,ipq->pq", lam, ih), prob.lagrangian_hessian(z, d["X0"][i], np.concatenate(
        [lam, np.zeros(prob.m - nh)]), 0.0), rtol=1e-11, atol=1e-12)
    assert np.all((np.abs(ih).sum(axis=0) != 0) <= (integ.hessianstructure() != 0))


@pytest.mark.parametrize("fwd", [True, False])
def test_rolling_model_level_layouts(fwd):
    """Model.forward / jacobian / hessian in the reference's [all x | all u] layout (tensorflow.py:236-340) against
    the oracle network on the gathered windows and the selector of gen_jac_proj_mat (jax.py:8-21)."""
    import pyneuralempc_amd as nEMPC
    nx, nu, w, H = 2, 1, 3, 5
    net = orc.MLP.random(w * (nx + nu), [24, 24], nx, seed=8)
    model = nEMPC.model.MLPModelRollingInput(net.W, net.b, nx, nu, rolling_window=w, forward_rolling=fwd,
                                             device="cuda:0")
    rng = np.random.default_rng(2)
    x, u = rng.normal(size=(H, nx)), rng.normal(size=(H, nu))
    px, pu = rng.normal(size=(w - 1, nx)), rng.normal(size=(w - 1, nu))
    model.set_prev_data(px, pu)
    xr, ur = orc.rolling_rows(x, u, px, pu, w, fwd)
    f, J, S = net.forward_jac_hess(np.concatenate([xr, ur], axis=1))
    np.testing.assert_allclose(model.forward(x, u), f, **F64)
    n = H * (nx + nu)
    P = np.zeros((n, H, w * (nx + nu)))
    for dim, vo, co in ((nx, 0, 0), (nu, H * nx, w * nx)):
        for i in range(H):
            for k in range(dim):
                for o in range(i, min(H, i + w)):
                    slot = (w - 1) - (o - i) if fwd else (o - i)
                    P[vo + i * dim + k, o, co + slot * dim + k] = 1.0
    np.testing.assert_allclose(model.jacobian(x, u), np.einsum("okc,voc->okv", J, P).reshape(H * nx, n), **F64)
    np.testing.assert_allclose(model.hessian(x, u), np.einsum("aoc,okcd,bod->okab", P, S, P), rtol=1e-11, atol=1e-12)
    with pytest.raises(AssertionError):
        model.set_prev_data(px[:1], pu)


def test_rolling_keras_dropin_and_rk4_rejection():
    import pyneuralempc_amd as nEMPC

    class _Layer:
        def __init__(self, w, b, act):
            self._w, self.activation = (w, b), act
        def get_weights(self):
            return list(self._w)

    def tanh(x): return x
    def linear(x): return x

    nx, nu, w = 2, 1, 2
    net = orc.MLP.random(w * (nx + nu), [16], nx, seed=0)

    class _Keras:
        layers = [_Layer(net.W[0], net.b[0], tanh), _Layer(net.W[1], net.b[1], linear)]
        input_shape, output_shape = (None, w * (nx + nu)), (None, nx)

    model = nEMPC.model.tensorflow.KerasTFModelRollingInput(_Keras(), nx, nu, rolling_window=w, device="cuda:0")
    with pytest.raises(NotImplementedError):
        nEMPC.integrator.rk4.RK4Integrator(model, 5, 0.1)
    with pytest.raises(ValueError, match="rolling"):
        nEMPC.model.tensorflow.KerasTFModelRollingInput(_Keras(), nx, nu, rolling_window=0, device="cuda:0")
    with pytest.raises(ValueError, match="input dim"):
        nEMPC.model.MLPModelRollingInput(net.W, net.b, nx, nu, rolling_window=3, device="cuda:0")


def test_nmpc_next_with_rolling_model_slsqp():
    """The reference's root test.py scenario (test.py:38-79: rolling model with window 2, DiscretIntegrator, H = 10,
    cost sum((u-2)^2), unbounded domain, NMPC.next from x_past) with a tanh MLP in place of its polynomial fake
    model: SLSQP on the CPU over the device callbacks; the result is feasible and a KKT point of the oracle's problem."""
    import warnings
    import pyneuralempc_amd as nEMPC
    nx, nu, w, H = 2, 1, 2, 10
    net = orc.MLP.random(w * (nx + nu), [24, 24], nx, seed=4)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    model = nEMPC.model.MLPModelRollingInput(net.W, net.b, nx, nu, rolling_window=w, forward_rolling=True,
                                             device="cuda:0")
    x_past, u_past = np.array([[0.2, 0.1]]), np.array([[0.0]])
    model.set_prev_data(x_past, u_past)
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.zeros((nx, nx)), R=np.eye(nu), uref=2.0, device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-np.inf, np.inf]] * nx,
                                             control_constraint=[[-np.inf, np.inf]])
    opt = nEMPC.optimizer.Slsqp(max_iteration=300, tolerance=1e-11, verbose=0)
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1, optimizer=opt)
    x0 = x_past.reshape(-1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pred, u = mpc.next(x0)
    assert pred is not None and pred.shape == (H, nx) and u.shape == (H, nu)
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=np.zeros((nx, nx)), R=np.eye(nu), uref=2.0, window=w,
                       hist_x=x_past, hist_u=u_past)
    z = np.concatenate([pred.ravel(), u.ravel()])
    assert np.abs(prob.constraints(z, x0)).max() < 1e-8
    # the cost does not see the states: the minimiser is u = 2 with the states following the dynamics
    np.testing.assert_allclose(u, 2.0, atol=1e-5)
    J, gr = prob.jacobian(z, x0), prob.gradient(z)
    lam = np.linalg.lstsq(J.T, -gr, rcond=None)[0]
    assert np.abs(gr + J.T @ lam).max() < 1e-5


@pytest.mark.parametrize("act", ["relu", "sigmoid", "softplus", "elu"])
def test_rolling_window_and_parameters_with_the_activation_family(act):
    """The activation is orthogonal to the input gather: a rolling window of 3 with time-varying / constant parameters
    (extra network inputs) and box rows, every activation, the three kernel families against the oracle incl. the
    Lagrangian Hessian and the Gauss-Newton callback."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, w, ne, B = 2, 1, 6, 3, 2, 5
    rng = np.random.default_rng(21)
    net = orc.MLP.random(w * (nx + nu) + ne, [32, 32], nx, seed=5, activations=act)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=8)
    hx, hu = rng.normal(size=(B, w - 1, nx)), rng.uniform(-1, 1, size=(B, w - 1, nu))
    ex = rng.normal(size=(H, ne))
    lamh, sigh, wh = rng.normal(size=(B, 2 * H * nx)), rng.uniform(0.5, 1.5, size=B), rng.uniform(0.2, 1.5, size=(B, H * nx))
    probs = [orc.Problem(net, H, nx, nu, orc.DISCRET, box=(-2.0, 2.0), extra=ex, window=w, hist_x=hx[i], hist_u=hu[i])
             for i in range(B)]
    g = np.stack([p.constraints(Zh[i], X0h[i]) for i, p in enumerate(probs)])
    jac = np.stack([p.jacobian(Zh[i], X0h[i]) for i, p in enumerate(probs)])
    hv = np.stack([p.hessian_values(Zh[i], X0h[i], lamh[i], sigh[i]) for i, p in enumerate(probs)])
    gn = np.stack([p.gauss_newton_values(Zh[i], X0h[i], wh[i], sigh[i]) for i, p in enumerate(probs)])
    for kernel in ("valu", "mfma", "mfma_tile"):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel,
                             n_extra=ne, rolling_window=w, activations=act)
        eng.set_box_rows(-2.0, 2.0)
        eng.bind_extra(eng.to_device(np.broadcast_to(ex[None], (B, H, ne)).copy()))
        eng.bind_history(eng.to_device(hx), eng.to_device(hu))
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        res = eng.eval(Z, X0, ("g", "jac_dense"))
        np.testing.assert_allclose(res["g"].cpu().numpy(), g, **F64)
        np.testing.assert_allclose(res["jac_dense"].cpu().numpy(), jac, **F64)
        h = eng.hess(Z, X0, eng.to_device(lamh), eng.to_device(sigh))["hvals"].cpu().numpy()
        np.testing.assert_allclose(h, hv, rtol=1e-10, atol=1e-11)
        hg = eng.hess_gn(Z, X0, eng.to_device(wh), eng.to_device(sigh))["hvals"].cpu().numpy()
        np.testing.assert_allclose(hg, gn, rtol=1e-11, atol=1e-12)


def test_nmpc_next_with_rolling_model_on_the_device_solver():
    """The same scenario with the batched device solver behind the Optimizer interface (DeviceSqp): the window is made
    the state inside nempc_solve; the answer is the SLSQP one."""
    import pyneuralempc_amd as nEMPC
    nx, nu, w, H = 2, 1, 2, 10
    net = orc.MLP.random(w * (nx + nu), [24, 24], nx, seed=4)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    model = nEMPC.model.MLPModelRollingInput(net.W, net.b, nx, nu, rolling_window=w, forward_rolling=True,
                                             device="cuda:0")
    x_past, u_past = np.array([[0.2, 0.1]]), np.array([[0.0]])
    model.set_prev_data(x_past, u_past)
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.zeros((nx, nx)), R=np.eye(nu), uref=2.0, device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-np.inf, np.inf]] * nx,
                                             control_constraint=[[-np.inf, np.inf]])
    opt = nEMPC.optimizer.DeviceSqp(max_iteration=200)
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1, optimizer=opt)
    x0 = x_past.reshape(-1)
    pred, u = mpc.next(x0)
    assert pred is not None and pred.shape == (H, nx) and u.shape == (H, nu)
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=np.zeros((nx, nx)), R=np.eye(nu), uref=2.0, window=w,
                       hist_x=x_past, hist_u=u_past)
    z = np.concatenate([pred.ravel(), u.ravel()])
    assert np.abs(prob.constraints(z, x0)).max() < 1e-8
    np.testing.assert_allclose(u, 2.0, atol=1e-6)


def test_next_batch_with_rolling_model_histories_and_rolled_parameters():
    """NMPC.next_batch on a rolling-window model with parameters: one history (states, controls, time-varying parameters)
    per problem, tvp rolled like the states.  Every problem against the oracle's problem with ITS history and extras."""
    import pyneuralempc_amd as nEMPC
    nx, nu, w, H, B, pd, td = 2, 1, 3, 8, 11, 1, 1
    net = orc.MLP.random(w * (nx + nu + td) + pd, [32, 32], nx, seed=9)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    model = nEMPC.model.MLPModelRollingInput(net.W, net.b, nx, nu, p_dim=pd, tvp_dim=td, rolling_window=w, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    Q, R = np.eye(nx), 0.1 * np.eye(nu)
    obj = nEMPC.objective.QuadraticObjective(Q=Q, R=R, device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-np.inf, np.inf]] * nx, control_constraint=[[-0.2, 0.2]])
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1, optimizer=nEMPC.optimizer.DeviceSqp())
    rng = np.random.default_rng(2)
    X0 = rng.uniform(-1, 1, size=(B, nx))
    px, pu = rng.uniform(-1, 1, size=(B, w - 1, nx)), rng.uniform(-0.2, 0.2, size=(B, w - 1, nu))
    ptv, tvp, p = rng.normal(size=(B, w - 1, td)), rng.normal(size=(B, H, td)), rng.normal(size=(B, pd))
    with pytest.raises(ValueError, match="prev_x"):
        mpc.next_batch(X0, p=p, tvp=tvp)                              # no history given, none stored
    xs, us, status = mpc.next_batch(X0, p=p, tvp=tvp, prev_x=px, prev_u=pu, prev_tvp=ptv, max_iter=300)
    assert (status == nEMPC.optimizer.Optimizer.SUCCESS).all()
    lb = np.concatenate([np.full(H * nx, -np.inf), np.full(H * nu, -0.2)])
    ub = -lb
    for i in range(B):
        model.set_prev_data(px[i], pu[i], ptv[i])
        ex = model.gather_extra(H, p=p[i], tvp=tvp[i])
        prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=Q, R=R, extra=ex, window=w, hist_x=px[i], hist_u=pu[i])
        z = np.concatenate([xs[i].ravel(), us[i].ravel()])
        assert np.abs(prob.constraints(z, X0[i])).max() < 1e-7
        assert (z >= lb - 1e-12).all() and (z <= ub + 1e-12).all()
        J, gr = prob.jacobian(z, X0[i]), prob.gradient(z)
        free = (z > lb + 1e-3) & (z < ub - 1e-3)
        lam = np.linalg.lstsq(J[:, free].T, -gr[free], rcond=None)[0]
        assert np.abs(gr[free] + J[:, free].T @ lam).max() < 1e-5 * max(1.0, np.abs(gr).max())
    # one stored history serves the whole batch when none is passed
    model.set_prev_data(px[0], pu[0], ptv[0])
    xs1, us1, st1 = mpc.next_batch(X0[:1], p=p[:1], tvp=tvp[:1], max_iter=300)
    np.testing.assert_allclose(us1[0], us[0], atol=1e-9)
