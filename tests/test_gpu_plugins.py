"""GPU: the reference-shaped plug-in surface (Model / Integrator / Objective / Constraint / Optimizer /
NMPC) driven like the reference's own scripts, checked against reference-generated goldens."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc
from helpers import GOLDEN, load_case, oracle_problem

pytestmark = pytest.mark.gpu
F64 = dict(rtol=1e-12, atol=1e-12)


def _integrator(d, W, b):
    import pyneuralempc_amd as nEMPC
    model = nEMPC.model.MLPModel(W, b, int(d["nx"]), int(d["nu"]), device="cuda:0")
    H, kind = int(d["H"]), int(d["kind"])
    if kind == orc.DISCRET:
        return nEMPC.integrator.discret.DiscretIntegrator(model, H)
    if kind == orc.UNITY:
        return nEMPC.integrator.unity.UnityIntegrator(model, H)
    return nEMPC.integrator.rk4.RK4Integrator(model, H, float(d["DT"]), cache_mode=True)


@pytest.mark.parametrize("name", ["c1_discret", "c2_unity", "c2_rk4", "odd_dims"])
def test_integrator_forward_jacobian_like_reference(name):
    d, W, b = load_case(name)
    integ = _integrator(d, W, b)
    H, nx, nu = int(d["H"]), int(d["nx"]), int(d["nu"])
    for i in range(d["Z"].shape[0]):
        z = d["Z"][i]
        states, u = z[:H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)
        np.testing.assert_allclose(integ.forward(states, u, d["X0"][i]), d["g_int"][i], **F64)
        np.testing.assert_allclose(integ.jacobian(states, u, d["X0"][i]), d["jac_int"][i], **F64)
    assert integ.get_lower_bounds(H) == [0.0] * (H * nx) and integ.get_upper_bounds(H) == [0.0] * (H * nx)
    with pytest.raises(AssertionError):
        integ.forward(states.ravel(), u, d["X0"][0])


def test_model_plugin_layouts():
    d, W, b = load_case("c1_discret")
    import pyneuralempc_amd as nEMPC
    model = nEMPC.model.MLPModel(W, b, 2, 1, device="cuda:0")
    net = orc.MLP(W, b)
    rng = np.random.default_rng(0)
    x, u = rng.normal(size=(6, 2)), rng.normal(size=(6, 1))
    xi = np.concatenate([x, u], axis=1)
    f, J, S = net.forward_jac_hess(xi)
    np.testing.assert_allclose(model.forward(x, u), f, **F64)
    MJ = model.jacobian(x, u)
    assert MJ.shape == (12, 18)
    Hs = model.hessian(x, u)
    assert Hs.shape == (6, 2, 18, 18)
    for t in range(6):
        np.testing.assert_allclose(MJ[2 * t:2 * t + 2, 2 * t:2 * t + 2], J[t][:, :2], **F64)
        np.testing.assert_allclose(MJ[2 * t:2 * t + 2, 12 + t:13 + t], J[t][:, 2:], **F64)
        np.testing.assert_allclose(Hs[t][:, 2 * t:2 * t + 2, 2 * t:2 * t + 2], S[t][:, :2, :2], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Hs[t][:, 12 + t, 2 * t:2 * t + 2], S[t][:, 2, :2], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Hs[t][:, 12 + t, 12 + t], S[t][:, 2, 2], rtol=1e-11, atol=1e-12)
    mask = np.ones((12, 18), dtype=bool)
    for t in range(6):
        mask[2 * t:2 * t + 2, 2 * t:2 * t + 2] = False
        mask[2 * t:2 * t + 2, 12 + t] = False
    assert not MJ[mask].any()


def test_torch_module_model_runs_on_the_kernels_and_matches_torch_autodiff():
    """TorchMLPModel: a torch.nn.Sequential dense network (BatchNorm1d, SiLU, GELU, a nested Sequential) read once and
    evaluated by the HIP kernels through the reference's Model surface -- forward / block-layout jacobian / hessian against
    torch's own autodiff of the module, and a closed-loop step through NMPC with the device optimizer."""
    import pyneuralempc_amd as nEMPC
    nn = torch.nn
    torch.manual_seed(1)
    bn = nn.BatchNorm1d(48)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    net = nn.Sequential(nn.Linear(3, 48), bn, nn.SiLU(), nn.Sequential(nn.Linear(48, 40), nn.GELU()), nn.Linear(40, 2)).double().eval()
    with torch.no_grad():
        net[-1].weight.mul_(0.2); net[-1].bias.mul_(0.2)
    for prm in net.parameters():
        prm.requires_grad_(False)
    model = nEMPC.model.TorchMLPModel(net, x_dim=2, u_dim=1, device="cuda:0")
    rng = np.random.default_rng(0)
    H = 5
    x, u = rng.normal(size=(H, 2)), rng.normal(size=(H, 1))
    xi = torch.tensor(np.concatenate([x, u], axis=1))
    np.testing.assert_allclose(model.forward(x, u), net(xi).detach().numpy(), rtol=1e-11, atol=1e-12)
    row = lambda v: net(v[None])[0]
    MJ, Hs = model.jacobian(x, u), model.hessian(x, u)
    for t in range(H):
        J = torch.func.jacrev(row)(xi[t]).detach().numpy()
        S = torch.func.hessian(row)(xi[t]).detach().numpy()
        np.testing.assert_allclose(MJ[2 * t:2 * t + 2, 2 * t:2 * t + 2], J[:, :2], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(MJ[2 * t:2 * t + 2, 2 * H + t], J[:, 2], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(Hs[t][:, 2 * t:2 * t + 2, 2 * t:2 * t + 2], S[:, :2, :2], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(Hs[t][:, 2 * H + t, 2 * H + t], S[:, 2, 2], rtol=1e-9, atol=1e-10)
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, 8)
    obj = nEMPC.objective.QuadraticObjective(Q=np.eye(2), R=0.1 * np.eye(1), device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-5.0, 5.0]] * 2, control_constraint=[[-1.0, 1.0]])
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], 8, 1.0, optimizer=nEMPC.optimizer.DeviceSqp())
    states, uu = mpc.next(np.array([0.4, -0.3]))
    assert states is not None and states.shape == (8, 2) and np.all(np.abs(uu) <= 1.0 + 1e-9)
    # the predicted states are the module's own roll-out of the controls
    xk = torch.tensor([0.4, -0.3], dtype=torch.float64)
    for t in range(8):
        xk = xk + net(torch.cat([xk, torch.tensor(uu[t])])[None])[0]
        np.testing.assert_allclose(states[t], xk.detach().numpy(), atol=1e-6)


def test_integrator_hessian_blocks_like_reference():
    d, W, b = load_case("c1_discret")
    integ = _integrator(d, W, b)
    H, nx, nu = 10, 2, 1
    z = d["Z"][0]
    states, u = z[:H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)
    ih = integ.hessian(states, u, d["X0"][0])
    assert ih.shape == (H * nx, 30, 30)
    prob = oracle_problem(d, W, b)
    contracted = np.einsum("i,ipq->pq", d["lam"][0][:H * nx], ih) + float(d["sigma"][0]) * prob.objective_hessian()
    np.testing.assert_allclose(contracted, d["hdense"][0], rtol=1e-11, atol=1e-12)
    S = integ.hessianstructure()
    assert np.all((np.abs(ih).sum(axis=0) != 0) <= (S != 0))


@pytest.mark.parametrize("name", ["c2_discret", "c5_box"])
def test_problem_glue_like_reference(name):
    """IpoptProblem / SlsqpProblem callbacks (one fused device evaluation per iterate)."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    from pyneuralempc_amd.optimizer.slsqp import SlsqpProblem
    d, W, b = load_case(name)
    integ = _integrator(d, W, b)
    obj = nEMPC.objective.QuadraticObjective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"],
                                             device="cuda:0")
    ctrs = [nEMPC.constraints.BoxStateConstraint(d["box_lo"], d["box_hi"])] if int(d["has_box"]) else []
    for i in range(d["Z"].shape[0]):
        pb = IpoptProblem(d["X0"][i], obj, ctrs, integ)
        assert pb._fused is not None
        z = d["Z"][i]
        np.testing.assert_allclose(pb.objective(z), d["f"][i], **F64)
        np.testing.assert_allclose(pb.gradient(z), d["grad"][i], **F64)
        np.testing.assert_allclose(pb.constraints(z), d["g"][i], **F64)
        np.testing.assert_allclose(pb.jacobian(z), d["jac"][i], **F64)
        assert pb._fused.n_device_evals == i + 1, "four callbacks of one iterate share one device evaluation"
        rows, cols = pb.hessianstructure()
        hv = pb.hessian(z, d["lam"][i], float(d["sigma"][i]))
        np.testing.assert_allclose(hv, d["hdense"][i][rows, cols], rtol=1e-11, atol=1e-12)
    np.testing.assert_array_equal(pb.get_constraint_lower_bounds(), d["cl"])
    np.testing.assert_array_equal(pb.get_constraint_upper_bounds(), d["cu_bound"])
    jr, jc = pb.jacobianstructure()
    assert np.array_equal(np.stack(np.nonzero(d["jac"][0] != 0)), np.stack([jr, jc]))
    sp = SlsqpProblem(d["X0"][0], obj, ctrs, integ)
    np.testing.assert_allclose(sp.constraints(d["Z"][0], eq=True), d["slsqp_eq"], **F64)
    np.testing.assert_allclose(sp.jacobian(d["Z"][0], eq=True), d["slsqp_eq_jac"], **F64)
    if ctrs:
        gi = sp.constraints(d["Z"][0], eq=False)
        st = d["Z"][0][:100]
        np.testing.assert_allclose(gi, np.concatenate([st + 2.0, 2.0 - st]), **F64)
        assert sp.jacobian(d["Z"][0], eq=False).shape == (200, 150)
        assert len(sp.get_constraints_dict()) == 2


def test_generic_plugin_path_matches_fused_path():
    """User-defined Objective / Constraint plug-ins (host callables) go through the unfused glue and
    must give the same numbers as the fused device path."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    d, W, b = load_case("c5_box")
    integ = _integrator(d, W, b)
    prob = oracle_problem(d, W, b)
    zcat = lambda s, u: np.concatenate([s.ravel(), u.ravel()])
    man = nEMPC.objective.ManualObjectifFunc(lambda s, u, p, t: prob.objective(zcat(s, u)),
                                             lambda s, u, p, t: prob.gradient(zcat(s, u)),
                                             lambda s, u, p, t: prob.objective_hessian())

    class MyBox(nEMPC.constraints.Constraint):      # not the recognised class -> unfused
        def forward(self, x, u, p=None, tvp=None): return x.reshape(-1).copy()
        def jacobian(self, x, u, p=None, tvp=None):
            return np.concatenate([np.eye(x.size), np.zeros((x.size, u.size))], axis=1)
        def get_dim(self, H): return 2 * H
        def get_lower_bounds(self, H): return np.full(2 * H, -2.0)
        def get_upper_bounds(self, H): return np.full(2 * H, 2.0)

    pb = IpoptProblem(d["X0"][0], man, [MyBox()], integ)
    assert pb._fused is None
    z = d["Z"][0]
    np.testing.assert_allclose(pb.objective(z), d["f"][0], **F64)
    np.testing.assert_allclose(pb.gradient(z), d["grad"][0], **F64)
    np.testing.assert_allclose(pb.constraints(z), d["g"][0], **F64)
    np.testing.assert_allclose(pb.jacobian(z), d["jac"][0], **F64)
    np.testing.assert_array_equal(pb.get_constraint_lower_bounds(), d["cl"])


def test_nmpc_next_slsqp_matches_reference_trajectory():
    """controller.NMPC.next end to end (SLSQP on the CPU drives the device callbacks) against the
    trajectory the reference produced with the same network, objective, bounds and options."""
    import pyneuralempc_amd as nEMPC
    m = dict(np.load(os.path.join(GOLDEN, "misc.npz")))
    W = [m[f"W{i}"] for i in range(3)]
    b = [m[f"b{i}"] for i in range(3)]
    H = int(m["nmpc_H"])
    model = nEMPC.model.MLPModel(W, b, 2, 1, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.eye(2), R=0.1 * np.eye(1), device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-5.0, 5.0]] * 2, control_constraint=[[-1.0, 1.0]])
    opt = nEMPC.optimizer.Slsqp(max_iteration=200, tolerance=1e-10, verbose=0, init_with_last_result=True)
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1.0, optimizer=opt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        states, u = mpc.next(m["nmpc_x0"])
        np.testing.assert_allclose(states, m["nmpc_states"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(u, m["nmpc_u"], rtol=1e-6, atol=1e-7)
        states2, u2 = mpc.next(m["nmpc_x1"])      # warm-started second solve
        np.testing.assert_allclose(states2, m["nmpc_states2"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(u2, m["nmpc_u2"], rtol=1e-6, atol=1e-7)
    assert np.abs(integ.forward(states, u, m["nmpc_x0"])).max() < 1e-8


def test_ipopt_optimizer_fails_loudly_without_cyipopt():
    import pyneuralempc_amd as nEMPC
    try:
        import cyipopt  # noqa: F401
        pytest.skip("cyipopt installed")
    except ImportError:
        pass
    d, W, b = load_case("c1_discret")
    integ = _integrator(d, W, b)
    obj = nEMPC.objective.QuadraticObjective(device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint([[-5, 5]] * 2, [[-1, 1]])
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], 10, 1.0)
    assert isinstance(mpc.optimizer, nEMPC.optimizer.Ipopt)
    with pytest.raises(ImportError, match="cyipopt"):
        mpc.next(d["X0"][0])


@pytest.mark.parametrize("name", ["tvp_p_discret", "tvp_p_rk4"])
def test_parameters_p_and_tvp_like_reference(name):
    """Constant (p) and time-varying (tvp) parameters are extra network inputs without Jacobian columns
    (model/tensorflow.py:39-47,65-66); they travel through Integrator / IpoptProblem exactly like in the reference."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    d, W, b = load_case(name)
    H, nx, nu = int(d["H"]), int(d["nx"]), int(d["nu"])
    model = nEMPC.model.MLPModel(W, b, nx, nu, p_dim=int(d["p_dim"]), tvp_dim=int(d["tvp_dim"]), device="cuda:0")
    integ = (nEMPC.integrator.discret.DiscretIntegrator(model, H) if int(d["kind"]) == orc.DISCRET
             else nEMPC.integrator.rk4.RK4Integrator(model, H, float(d["DT"])))
    obj = nEMPC.objective.QuadraticObjective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"],
                                             device="cuda:0")
    p, tvp = d["p"], d["tvp"]
    for i in range(d["Z"].shape[0]):
        z = d["Z"][i]
        states, u = z[:H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)
        np.testing.assert_allclose(integ.forward(states, u, d["X0"][i], p=p, tvp=tvp), d["g_int"][i], **F64)
        np.testing.assert_allclose(integ.jacobian(states, u, d["X0"][i], p=p, tvp=tvp), d["jac_int"][i], **F64)
        pb = IpoptProblem(d["X0"][i], obj, [], integ, p=p, tvp=tvp)
        assert pb._fused is not None
        np.testing.assert_allclose(pb.constraints(z), d["g"][i], **F64)
        np.testing.assert_allclose(pb.jacobian(z), d["jac"][i], **F64)
        rows, cols = pb.hessianstructure()
        np.testing.assert_allclose(pb.hessian(z, d["lam"][i], float(d["sigma"][i])), d["hdense"][i][rows, cols],
                                   rtol=1e-11, atol=1e-12)
    with pytest.raises(ValueError):
        integ.forward(states, u, d["X0"][0])            # parameters missing
    # Model plug-in: forward with parameters against the oracle network
    net = orc.MLP(W, b)
    ex = np.concatenate([tvp, np.tile(p.reshape(1, -1), (H, 1))], axis=1)
    np.testing.assert_allclose(model.forward(states, u, p=p, tvp=tvp), net.forward(np.concatenate([states, u, ex], axis=1)), **F64)


def test_lotka_volterra_example_runs():
    """examples/lotka_volterra/run.py (counterpart of the reference's example script): a few closed-loop steps with
    SLSQP on the device callbacks, and the batched on-device solve of the same problem family."""
    import importlib.util
    path = os.path.join(os.path.dirname(GOLDEN), "..", "examples", "lotka_volterra", "run.py")
    spec = importlib.util.spec_from_file_location("lv_example", os.path.abspath(path))
    lv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lv)
    traj = lv.main(steps=3, fit_iters=300, verbose=False)
    assert traj.shape == (4, 2) and np.all(np.isfinite(traj))
    assert traj[:, 0].max() <= lv.X_MAX + 0.05          # state limit respected up to the surrogate's model error
    X = lv.main(steps=2, batch=16, fit_iters=300, verbose=False)
    assert X.shape == (16, 2) and np.all(np.isfinite(X))
    # the same closed loop with the whole solve on the device (optimizer.DeviceSqp behind NMPC.next)
    traj_dev = lv.main(steps=3, fit_iters=300, verbose=False, device_solver=True)
    assert traj_dev.shape == (4, 2) and np.all(np.isfinite(traj_dev))
    assert traj_dev[:, 0].max() <= lv.X_MAX + 0.05
    # (an economic cost, linear in u, on a surrogate refitted per call: the two solvers need not pick the same minimiser
    # step by step -- the plant paths stay within a few 1e-2 of each other over these steps)
    np.testing.assert_allclose(traj_dev, traj, atol=0.1)


def test_torch_objective_through_the_unfused_glue():
    """A user-supplied differentiable cost (TorchObjectifFunc, the JAXObjectifFunc counterpart) with the device
    integrator: same numbers as the fused quadratic path and the reference golden."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    d, W, b = load_case("c2_discret")
    integ = _integrator(d, W, b)
    dev = torch.device("cuda:0")
    Qt, Rt, xr, ur, ct = (torch.tensor(d[k], device=dev) for k in ("Q", "R", "xref", "uref", "cu"))

    def cost(states, u, p=None, tvp=None):
        dx, du = states - xr, u - ur
        return torch.einsum("ti,ij,tj->", dx, Qt, dx) + torch.einsum("ti,ij,tj->", du, Rt, du) + torch.sum(ct * u)

    obj = nEMPC.objective.TorchObjectifFunc(cost, device="cuda:0")
    pb = IpoptProblem(d["X0"][0], obj, [], integ)
    assert pb._fused is None
    z = d["Z"][0]
    np.testing.assert_allclose(pb.objective(z), d["f"][0], rtol=1e-12)
    np.testing.assert_allclose(pb.gradient(z), d["grad"][0], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(pb.constraints(z), d["g"][0], **F64)
    np.testing.assert_allclose(pb.jacobian(z), d["jac"][0], **F64)
    np.random.seed(11)
    rows, cols = pb.hessianstructure()
    hv = pb.hessian(z, d["lam"][0], float(d["sigma"][0]))
    np.testing.assert_allclose(hv, d["hdense"][0][rows, cols], rtol=1e-10, atol=1e-11)


def test_b1_fast_path_and_sparse_view_match_the_engine():
    """The B=1 callback path (pinned, device-mapped staging; optimizer/base.py:_FusedEvaluator) returns what a plain
    batched evaluation returns, for the dense and for the sparse-Jacobian view (band values straight from the device:
    Ipopt(sparse_jacobian=True) never builds or copies the dense matrix), incl. the Hessian callback."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem, _SparseJacobianView
    d, W, b = load_case("c5_box")
    H, nx, nu = int(d["H"]), int(d["nx"]), int(d["nu"])
    model = nEMPC.model.MLPModel(W, b, nx, nu, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"], device="cuda:0")
    box = nEMPC.constraints.BoxStateConstraint(d["box_lo"], d["box_hi"])
    for i in range(d["Z"].shape[0]):
        z, x0 = d["Z"][i], d["X0"][i]
        pb = IpoptProblem(x0, obj, [box], integ)
        assert pb._fused is not None
        n_before = pb._fused.n_device_evals
        np.testing.assert_allclose(pb.objective(z), d["f"][i], **F64)
        np.testing.assert_allclose(pb.gradient(z), d["grad"][i], **F64)
        np.testing.assert_allclose(pb.constraints(z), d["g"][i], **F64)
        np.testing.assert_allclose(pb.jacobian(z), d["jac"][i], **F64)
        assert pb._fused.n_device_evals == n_before + 1             # four callbacks, one device evaluation
        sv = _SparseJacobianView(pb, True)
        rows, cols = sv.jacobianstructure()
        vals = sv.jacobian(z)
        assert np.array_equal(vals, d["jac"][i][rows, cols]) or np.allclose(vals, d["jac"][i][rows, cols], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sv.constraints(z), d["g"][i], **F64)
        np.testing.assert_allclose(sv.objective(z), d["f"][i], **F64)
        hr, hc = pb.hessianstructure()
        np.testing.assert_allclose(pb.hessian(z, d["lam"][i], float(d["sigma"][i])), d["hdense"][i][hr, hc], rtol=1e-11, atol=1e-12)


def test_waiter_path_matches_the_synchronised_path_bit_for_bit():
    """The B=1 callback path reads its results from coherent host memory as soon as the stream's write-value word flips
    (optimizer/base.py:_StreamWaiter; the visibility contract is stated there).  Backstop of that argument: a few hundred
    callbacks of alternating iterates, dense and sparse view and the Hessian, the result buffer poisoned before every call,
    against the same call followed by stream.synchronize()."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.base import _StreamWaiter
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    d, W, b = load_case("c5_box")
    H, nx, nu = int(d["H"]), int(d["nx"]), int(d["nu"])
    model = nEMPC.model.MLPModel(W, b, nx, nu, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"], device="cuda:0")
    box = nEMPC.constraints.BoxStateConstraint(d["box_lo"], d["box_hi"])
    pb = IpoptProblem(d["X0"][0], obj, [box], integ)
    fe = pb._fused
    rng = np.random.default_rng(0)
    iterates = [d["Z"][i % d["Z"].shape[0]] + 0.01 * rng.normal(size=d["Z"].shape[1]) for i in range(6)]
    x0s = [d["X0"][i % d["X0"].shape[0]] for i in range(6)]
    lam = rng.normal(size=fe.engine.m)

    class Sync:                                     # the reference behaviour: drain the stream
        def __init__(self, stream): self.stream = stream
        def wait(self): self.stream.synchronize()

    def run(sparse, use_sync, reps):
        out = []
        st = fe._fast(sparse)
        keep = st["waiter"]
        assert isinstance(keep, _StreamWaiter) and keep.flag is not None     # the write-value path is the one under test
        if use_sync:
            st["waiter"] = Sync(st["stream"])
        try:
            for k in range(reps):
                st["out"][:] = np.nan                # poison: a result read before it has landed cannot pass
                fe._key = None
                v = fe.evaluate(iterates[k % 6], x0s[k % 6], sparse=sparse)
                out.append(np.concatenate([[v["f"]], v["grad"], v["g"], np.ravel(v["jac_sparse" if sparse else "jac_dense"])]))
        finally:
            st["waiter"] = keep
        return out
    for sparse in (False, True):
        ref = run(sparse, True, 12)
        got = run(sparse, False, 300)
        for k, v in enumerate(got):
            assert not np.isnan(v).any()
            assert np.array_equal(v, ref[k % 6]), (sparse, k)
    hs_ref = None
    for k in range(200):
        if fe._hess_state is not None:
            fe._hess_state["out"][:] = np.nan
        hv = fe.hessian_values(iterates[k % 6], x0s[k % 6], lam, 1.0 + (k % 6))
        assert not np.isnan(hv).any()
        if k < 6:
            hs_ref = (hs_ref or []) + [hv]
        else:
            assert np.array_equal(hv, hs_ref[k % 6]), k


@pytest.mark.parametrize("kind,DT", [("discret", 1.0), ("unity", 1.0), ("rk4", 0.2)])
def test_torch_model_matches_the_kernel_path(kind, DT):
    """The same network expressed both ways -- MLPModel (HIP kernels) and model.TorchModel (a torch callable on the
    device, differentiated by torch.func, through the integrators' host algebra) -- gives the same defects, Jacobian,
    Hessian rows and, through SLSQP, the same NMPC step."""
    import pyneuralempc_amd as nEMPC
    m = dict(np.load(os.path.join(GOLDEN, "misc.npz")))
    W = [m[f"W{i}"] for i in range(3)]
    b = [m[f"b{i}"] for i in range(3)]
    H = 6
    Wt = [torch.tensor(w, device="cuda:0") for w in W]
    bt = [torch.tensor(x, device="cuda:0") for x in b]

    def f(x, u, p=None, tvp=None):
        a = torch.cat([x, u], dim=-1)
        for w, bb in zip(Wt[:-1], bt[:-1]):
            a = torch.tanh(a @ w + bb)
        return a @ Wt[-1] + bt[-1]
    dev_model = nEMPC.model.MLPModel(W, b, 2, 1, device="cuda:0")
    mk = {"discret": lambda mod: nEMPC.integrator.DiscretIntegrator(mod, H), "unity": lambda mod: nEMPC.integrator.UnityIntegrator(mod, H),
          "rk4": lambda mod: nEMPC.integrator.RK4Integrator(mod, H, DT)}[kind]
    rng = np.random.default_rng(4)
    x, u, x0 = rng.normal(size=(H, 2)), rng.uniform(-1, 1, size=(H, 1)), rng.uniform(-1, 1, size=2)
    ref = mk(dev_model)
    for vector_mode in (True, False):
        tm = nEMPC.model.TorchModel(f, 2, 1, vector_mode=vector_mode, device="cuda:0")
        integ = mk(tm)
        np.testing.assert_allclose(integ.forward(x, u, x0), ref.forward(x, u, x0), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(integ.jacobian(x, u, x0), ref.jacobian(x, u, x0), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(integ.hessian(x, u, x0), ref.hessian(x, u, x0), rtol=1e-9, atol=1e-10)
    if kind == "discret":
        obj = nEMPC.objective.QuadraticObjective(Q=np.eye(2), R=0.1 * np.eye(1), device="cuda:0")
        out = {}
        for name, model in (("kernel", dev_model), ("torch", nEMPC.model.TorchModel(f, 2, 1, vector_mode=False, device="cuda:0"))):
            dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-5.0, 5.0]] * 2, control_constraint=[[-1.0, 1.0]])
            mpc = nEMPC.controller.NMPC(nEMPC.integrator.DiscretIntegrator(model, H), obj, [dom], H, 1.0,
                                        optimizer=nEMPC.optimizer.Slsqp(verbose=0, tolerance=1e-10))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[name] = mpc.next(m["nmpc_x0"])
            assert out[name][0] is not None
        np.testing.assert_allclose(out["torch"][0], out["kernel"][0], atol=1e-6)
        np.testing.assert_allclose(out["torch"][1], out["kernel"][1], atol=1e-6)
        with pytest.raises(NotImplementedError, match="next_batch"):
            mpc.next_batch(np.stack([m["nmpc_x0"]]))
