"""GPU: the batched on-device solver (Gauss-Newton SQP + Riccati + log barrier) against the trajectory the
reference produced with its SLSQP optimizer, and against SciPy SLSQP driven by the CPU oracle."""
import os
import warnings

import numpy as np
import pytest
import torch
from scipy.optimize import Bounds, minimize

from oracle import nempc_oracle as orc
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _slsqp_oracle(prob, x0, lb, ub):
    """Checker: the same NLP solved on the CPU with SciPy SLSQP on the oracle's callbacks."""
    zi = orc.cold_start(x0, prob.H, prob.nu)
    zi = np.clip(zi, np.where(np.isfinite(lb), lb + 1e-3, -np.inf), np.where(np.isfinite(ub), ub - 1e-3, np.inf))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize(prob.objective, zi, method="SLSQP", jac=prob.gradient, bounds=Bounds(lb, ub),
                       constraints=[{"type": "eq", "fun": lambda z: prob.constraints(z, x0),
                                     "jac": lambda z: prob.jacobian(z, x0)}],
                       options={"maxiter": 500, "ftol": 1e-12})
    return res


def test_next_batch_matches_reference_slsqp_trajectory():
    import pyneuralempc_amd as nEMPC
    m = dict(np.load(os.path.join(GOLDEN, "misc.npz")))
    W = [m[f"W{i}"] for i in range(3)]
    b = [m[f"b{i}"] for i in range(3)]
    H = int(m["nmpc_H"])
    model = nEMPC.model.MLPModel(W, b, 2, 1, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.eye(2), R=0.1 * np.eye(1), device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-5.0, 5.0]] * 2, control_constraint=[[-1.0, 1.0]])
    mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1.0, optimizer=nEMPC.optimizer.Slsqp())
    X0 = np.stack([m["nmpc_x0"], m["nmpc_x1"]])
    states, u, status = mpc.next_batch(X0)
    assert status.tolist() == [0, 0]
    np.testing.assert_allclose(states[0], m["nmpc_states"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(u[0], m["nmpc_u"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(states[1], m["nmpc_states2"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(u[1], m["nmpc_u2"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", ["unbounded", "active_control_bounds", "rk4_6x3"])
def test_batched_solve_matches_scipy_on_the_oracle(case):
    from pyneuralempc_amd import CallbackEngine
    if case == "rk4_6x3":
        nx, nu, hidden, H, kind, DT, B = 6, 3, [32, 32], 8, orc.RK4, 0.1, 12
    else:
        nx, nu, hidden, H, kind, DT, B = 2, 1, [64, 64], 20, orc.DISCRET, 1.0, 48
    net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    prob = orc.Problem(net, H, nx, nu, kind, DT, Q=np.eye(nx), R=0.1 * np.eye(nu))
    n = prob.n
    lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    if case == "active_control_bounds":
        lb[H * nx:], ub[H * nx:] = -0.05, 0.05       # tight: most controls end on a bound
        lb[:H * nx], ub[:H * nx] = -3.0, 3.0
    elif case == "rk4_6x3":
        lb[H * nx:], ub[H * nx:] = -0.5, 0.5
    X0 = np.random.default_rng(11).uniform(-1.0, 1.0, size=(B, nx))
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator={0: "discret", 2: "rk4"}[kind], DT=DT, dtype=torch.float64,
                         device="cuda:0", max_batch=B)
    eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
    # (an outer iteration carries one trial evaluation under deferred backtracking: a rejected step costs an iteration)
    Z, status, iters = eng.solve(eng.to_device(X0), lb=lb, ub=ub, max_iter=400)
    Z, status = Z.cpu().numpy(), status.cpu().numpy()
    assert (status == 0).all(), f"{int((status != 0).sum())} problems did not converge in {iters} iterations"
    for i in range(B):
        assert np.abs(prob.constraints(Z[i], X0[i])).max() < 1e-7
        assert (Z[i] >= lb - 1e-12).all() and (Z[i] <= ub + 1e-12).all()
    same = 0
    for i in range(0, B, max(1, B // 6)):
        # first-order optimality at the returned point (oracle derivatives): stationarity on the free variables
        J, gr = prob.jacobian(Z[i], X0[i]), prob.gradient(Z[i])
        free = (Z[i] > lb + 1e-3) & (Z[i] < ub - 1e-3)   # clear of the bounds: barrier gradient mu/d <= 1e-6
        lam = np.linalg.lstsq(J[:, free].T, -gr[free], rcond=None)[0]
        assert np.abs(gr[free] + J[:, free].T @ lam).max() < 1e-5 * max(1.0, np.abs(gr).max())
        rest = gr + J.T @ lam                 # on a bound the reduced gradient must push into the bound
        assert (rest[Z[i] <= lb + 1e-3] > -1e-3).all() and (rest[Z[i] >= ub - 1e-3] < 1e-3).all()
        # the NLP is non-convex: SLSQP may stop in another local minimum, but never in a better one
        ref = _slsqp_oracle(prob, X0[i], lb, ub)
        f_gpu = prob.objective(Z[i])
        assert f_gpu <= ref.fun * (1 + 1e-6) + 1e-8, (f_gpu, ref.fun)
        if abs(f_gpu - ref.fun) <= 1e-5 * max(1.0, abs(ref.fun)):
            np.testing.assert_allclose(Z[i], ref.x, rtol=0, atol=2e-4)
            same += 1
    assert same >= 3, "most sampled problems should land in SLSQP's minimum"


def test_solver_argument_errors():
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd._lib import NempcError
    net = orc.MLP.random(3, [16], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, 4, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=2)
    X0 = eng.to_device(np.zeros((2, 2)))
    with pytest.raises(NempcError):
        eng.solve(X0, lb=np.ones(eng.n), ub=np.zeros(eng.n))        # lb > ub
    with pytest.raises(NempcError):
        eng.solve(X0, mu_factor=1.5)
    with pytest.raises(NempcError, match="fixed variable"):
        eng.solve(X0, lb=np.zeros(eng.n), ub=np.concatenate([np.zeros(1), np.ones(eng.n - 1)]))   # lb == ub


def test_next_batch_turns_box_state_rows_into_state_bounds():
    """BoxStateConstraint rows (BASELINE config 5 style) are bounds on the state variables for the batched solver:
    same solution as the identical limits given through the DomainConstraint; other row constraints are refused."""
    import pyneuralempc_amd as nEMPC
    nx, nu, H, B = 2, 1, 12, 16
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=2)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    X0 = np.random.default_rng(3).uniform(-0.6, 0.6, size=(B, nx))

    def controller(state_lim, extra):
        model = nEMPC.model.MLPModel(net.W, net.b, nx, nu, device="cuda:0")
        integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
        obj = nEMPC.objective.QuadraticObjective(Q=np.eye(nx), R=0.05 * np.eye(nu), xref=np.full((H, nx), 1.0),
                                                 device="cuda:0")
        dom = nEMPC.constraints.DomainConstraint(states_constraint=[state_lim] * nx, control_constraint=[[-1.0, 1.0]])
        return nEMPC.controller.NMPC(integ, obj, [dom] + extra, H, 1.0, optimizer=nEMPC.optimizer.Slsqp())

    a = controller([-0.7, 0.7], [])
    b = controller([-5.0, 5.0], [nEMPC.constraints.BoxStateConstraint(-0.7, 0.7, x_dim=nx)])
    sa, ua, sta = a.next_batch(X0, max_iter=80)
    sb, ub_, stb = b.next_batch(X0, max_iter=80)
    assert (sta == 0).sum() >= B // 2 and np.array_equal(sta, stb)
    assert np.array_equal(sa, sb) and np.array_equal(ua, ub_)
    ok = sta == 0
    assert sa[ok].max() <= 0.7 + 1e-9 and sa[ok].max() > 0.69  # the limit is active (xref = 1 pulls the states up)

    class Rows(nEMPC.constraints.InequalityConstraint):
        def get_dim(self, H): return H
    with pytest.raises(NotImplementedError):
        controller([-5.0, 5.0], [Rows()]).next_batch(X0)


@pytest.mark.parametrize("dims", [(2, 1, [32, 32], 10, orc.DISCRET, 1.0), (3, 2, [24], 7, orc.DISCRET, 1.0),
                                  (6, 3, [32, 32], 8, orc.RK4, 0.1)])
def test_riccati_sweep_per_thread_and_per_wave_agree(dims):
    """The two LQ kernels evaluate the same recursion in the same order: identical iterates, statuses and iteration
    counts on bounded problems (compile-time 2/1 and 6/3 instantiations and the runtime-dimension one)."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, kind, DT = dims
    B = 9
    net = orc.MLP.random(nx + nu, hidden, nx, seed=5)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator={0: "discret", 2: "rk4"}[kind], DT=DT, device="cuda:0",
                         max_batch=B)
    X0 = eng.to_device(np.random.default_rng(2).uniform(-0.8, 0.8, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -2.0), np.full(H * nu, -0.3)])
    out = {k: eng.solve(X0, lb=lb, ub=-lb, max_iter=60, lq_kernel=k) for k in ("thread", "wave", "auto")}
    Zt, st, it = out["thread"]
    Zw, sw, iw = out["wave"]
    assert it == iw and torch.equal(st, sw)
    np.testing.assert_allclose(Zw.cpu().numpy(), Zt.cpu().numpy(), rtol=0, atol=1e-12)
    assert torch.equal(out["auto"][0], Zw if nx * (nx + nu) >= 12 else Zt)
    assert int((st == 0).sum()) >= B - 2


def test_next_batch_with_p_and_tvp_binds_per_problem_parameters():
    """Models with constant / time-varying parameters: next_batch takes p / tvp per problem (or one set for all) and
    never evaluates B problems against the (1, H, n_extra) binding a previous NMPC.next left behind."""
    import pyneuralempc_amd as nEMPC
    nx, nu, H, B, pd, td = 2, 1, 8, 5, 1, 2
    net = orc.MLP.random(nx + nu + td + pd, [32, 32], nx, seed=5)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    rng = np.random.default_rng(8)
    X0 = rng.uniform(-0.5, 0.5, size=(B, nx))
    P = rng.normal(size=(B, pd))
    TV = rng.normal(size=(B, H, td))
    model = nEMPC.model.MLPModel(net.W, net.b, nx, nu, p_dim=pd, tvp_dim=td, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.eye(nx), R=0.05 * np.eye(nu), xref=np.full((H, nx), 0.3), device="cuda:0")
    dom = nEMPC.constraints.DomainConstraint(states_constraint=[[-3.0, 3.0]] * nx, control_constraint=[[-1.0, 1.0]])
    ctl = nEMPC.controller.NMPC(integ, obj, [dom], H, 1.0, optimizer=nEMPC.optimizer.Slsqp())
    # a single-problem solve first: leaves a one-problem parameter binding on the shared handle
    s1, u1 = ctl.next(X0[0], p=P[0], tvp=TV[0])
    assert s1 is not None
    with pytest.raises(ValueError, match="p_dim"):
        ctl.next_batch(X0, tvp=TV)
    S, U, st = ctl.next_batch(X0, p=P, tvp=TV, max_iter=80)
    assert (st == 0).all()
    # every problem solved with ITS parameters: the defects of the returned trajectories vanish under the oracle
    for b in range(B):
        extra = np.concatenate([TV[b], np.tile(P[b].reshape(1, -1), (H, 1))], axis=1)
        prob = orc.Problem(net, H, nx, nu, extra=extra)
        z = np.concatenate([S[b].ravel(), U[b].ravel()])
        assert np.abs(prob.constraints(z, X0[b])).max() < 1e-7
    # problem 0 agrees with the single-problem SLSQP solve of the same data
    np.testing.assert_allclose(S[0], s1, atol=5e-4)   # SLSQP stops at its own ftol
    # one parameter set for all problems == that set repeated
    Sa, Ua, _ = ctl.next_batch(X0, p=P[1], tvp=TV[1], max_iter=80)
    Sb, Ub, _ = ctl.next_batch(X0, p=np.tile(P[1], (B, 1)), tvp=np.tile(TV[1][None], (B, 1, 1)), max_iter=80)
    assert np.array_equal(Sa, Sb) and np.array_equal(Ua, Ub)
    # inner-loop backtracking gathers the still-searching problems (their parameters with them) for its later trials: the
    # same solutions as the deferred schedule, and the defects vanish under each problem's own parameters
    Sl, Ul, stl = ctl.next_batch(X0, p=P, tvp=TV, max_iter=80, linesearch="loop")
    assert (stl == 0).all()
    np.testing.assert_allclose(Sl, S, atol=1e-5)
    np.testing.assert_allclose(Ul, U, atol=1e-5)
    for b in range(B):
        extra = np.concatenate([TV[b], np.tile(P[b].reshape(1, -1), (H, 1))], axis=1)
        z = np.concatenate([Sl[b].ravel(), Ul[b].ravel()])
        assert np.abs(orc.Problem(net, H, nx, nu, extra=extra).constraints(z, X0[b])).max() < 1e-7


def test_objective_edits_between_solves_reach_the_device():
    """QuadraticObjective.params edited after the first solve (a moving reference in tracking MPC): the fused
    evaluator, the objective's own engine and next_batch all see the new values (the cache was keyed by id()
    and uploaded once)."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    nx, nu, H = 2, 1, 6
    net = orc.MLP.random(nx + nu, [16], nx, seed=1)
    model = nEMPC.model.MLPModel(net.W, net.b, nx, nu, device="cuda:0")
    integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
    obj = nEMPC.objective.QuadraticObjective(Q=np.eye(nx), R=0.1 * np.eye(nu), xref=np.zeros((H, nx)), device="cuda:0")
    box = nEMPC.constraints.BoxStateConstraint(-1.0, 1.0, x_dim=nx)
    Z, X0 = orc.synthetic_inputs(1, H, nx, nu, seed=3)
    z, x0 = Z[0], X0[0]
    pb = IpoptProblem(x0, obj, [box], integ)
    f0 = pb.objective(z)
    np.testing.assert_allclose(f0, orc.Problem(net, H, nx, nu).objective(z), rtol=1e-12)
    obj.params["xref"] = np.full((H, nx), 0.5)
    obj.params["QT"] = 3.0 * np.eye(nx)
    ref = orc.Problem(net, H, nx, nu, xref=np.full((H, nx), 0.5), QT=3.0 * np.eye(nx))
    pb2 = IpoptProblem(x0, obj, [box], integ)
    assert pb2._fused is pb._fused                      # same handle, refreshed parameters
    np.testing.assert_allclose(pb2.objective(z), ref.objective(z), rtol=1e-12)
    np.testing.assert_allclose(pb2.gradient(z), ref.gradient(z), rtol=1e-12, atol=1e-12)
    st, u = z[:H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)
    np.testing.assert_allclose(obj.forward(st, u), ref.objective(z), rtol=1e-12)
    np.testing.assert_allclose(obj.hessian(st, u), ref.objective_hessian(), rtol=1e-12)
    # the box bounds are re-read as well
    box.lo, box.hi = np.full(nx, -0.25), np.full(nx, 0.25)
    pb3 = IpoptProblem(x0, obj, [box], integ)
    assert np.allclose(pb3.get_constraint_lower_bounds()[H * nx:], -0.25)
    # the evaluator cache is bounded
    for k in range(8):
        o = nEMPC.objective.QuadraticObjective(xref=np.full((H, nx), 0.1 * k), device="cuda:0")
        IpoptProblem(x0, o, [], integ)
    assert len(integ._fused) <= 4


@pytest.mark.parametrize("dims", [(2, 1, [64, 64], 20, "discret", 1.0, torch.float64), (6, 3, [32, 32], 8, "rk4", 0.1, torch.float32)])
def test_compaction_gives_the_same_solutions_and_reports_iterations(dims):
    """nempc_solver_opts.compact: gathering the unconverged problems to the front changes which slot a problem sits in,
    nothing else -- solutions, statuses and per-problem iteration counts are identical to the lock-step run."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, kind, DT, dtype = dims
    B = 300
    net = orc.MLP.random(nx + nu, hidden, nx, seed=3)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=kind, DT=DT, dtype=dtype, device="cuda:0", max_batch=B)
    X0 = eng.to_device(np.random.default_rng(4).uniform(-0.8, 0.8, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    Za, sa, ia, pa = eng.solve(X0, lb=lb, ub=-lb, max_iter=60, compact=False, return_iterations=True)
    Zb, sb, ib, pb = eng.solve(X0, lb=lb, ub=-lb, max_iter=60, compact=True, return_iterations=True)
    assert torch.equal(sa, sb) and torch.equal(pa, pb) and ia == ib
    assert torch.equal(Za, Zb)
    ok = sa == 0
    assert int(ok.sum()) >= B // 2
    assert int(pa[ok].min()) >= 1 and int(pa[ok].max()) <= ia and int(pa[~ok].max() if (~ok).any() else 0) == 0
    # spread of convergence iterations: compaction has something to do
    assert int(pa[ok].max()) > int(pa[ok].min())


def test_box_rows_are_state_bounds_for_the_batched_solver():
    """nempc_solve with box ROWS enabled on the handle (BASELINE configs[4]): the rows' limits are intersected with the
    state bounds inside the library -- same iterates as passing them as lb / ub on a handle without box rows."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 12, 40
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=2)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    X0h = np.random.default_rng(3).uniform(-0.6, 0.6, size=(B, nx))
    xref = np.full((H, nx), 1.0)
    plain = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    boxed = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    for e in (plain, boxed):
        e.set_objective(Q=np.eye(nx), R=0.05 * np.eye(nu), xref=xref)
    boxed.set_box_rows(-0.7, 0.7)
    assert boxed.m == 2 * H * nx
    wide = np.concatenate([np.full(H * nx, -5.0), np.full(H * nu, -1.0)])
    tight = np.concatenate([np.full(H * nx, -0.7), np.full(H * nu, -1.0)])
    Za, sa, _ = plain.solve(plain.to_device(X0h), lb=tight, ub=-tight, max_iter=80)
    Zb, sb, _ = boxed.solve(boxed.to_device(X0h), lb=wide, ub=-wide, max_iter=80)
    assert torch.equal(sa, sb) and torch.equal(Za, Zb)
    ok = (sa == 0).cpu().numpy()
    assert ok.sum() >= B // 2
    zs = Za.cpu().numpy()[ok][:, :H * nx]
    assert zs.max() <= 0.7 + 1e-9 and zs.max() > 0.69        # the row limit is active


def test_backtracking_modes_reach_the_same_solutions_and_barrier_kinds_agree():
    """nempc_solver_opts.linesearch / .barrier: deferred backtracking (one trial evaluation per iteration; a rejected
    problem retries at half the length next iteration) and the inner loop are different schedules of the same search --
    on problems both converge, to the same KKT points; the primal-dual interior point and the primal log barrier end at
    the same bounded minimisers.  Deferral may spend more iterations, never fewer converged problems at a generous budget."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 20, 256
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    out = {}
    for key, kw in (("loop", dict(linesearch="loop")), ("deferred", dict(linesearch="deferred")),
                    ("primal", dict(linesearch="loop", barrier="primal"))):
        Z, st, it, per = eng.solve(X0, lb=lb, ub=-lb, max_iter=400, return_iterations=True, **kw)
        out[key] = (Z.cpu().numpy(), st.cpu().numpy(), it, per.cpu().numpy())
    ok_l, ok_d, ok_p = (out[k][1] == 0 for k in ("loop", "deferred", "primal"))
    assert ok_l.mean() > 0.97 and ok_d.mean() > 0.97 and ok_p.mean() > 0.9
    both = ok_l & ok_d
    # same local solution unless the two schedules fell into different basins (rare; bounded here)
    diff = np.abs(out["loop"][0][both] - out["deferred"][0][both]).max(axis=1)
    assert (diff < 1e-5).mean() > 0.97
    bothp = ok_l & ok_p
    diffp = np.abs(out["loop"][0][bothp] - out["primal"][0][bothp]).max(axis=1)
    assert (diffp < 1e-4).mean() > 0.95
    # bounds hold at every returned iterate
    for k in out:
        assert (out[k][0] >= lb - 1e-12).all() and (out[k][0] <= -lb + 1e-12).all()
    # a deferred problem spends an iteration per trial
    assert np.median(out["deferred"][3][both]) >= np.median(out["loop"][3][both])


def test_iteration_shortcuts_do_not_change_the_iterates(monkeypatch):
    """Three things the solver does to shorten an iteration are schedules, not approximations -- each can be switched off
    by an environment knob and must give bit-identical iterates, statuses and iteration counts: the damping levels of a
    Riccati sweep tried side by side on lanes of one wave (NEMPC_LQ_SPEC=1: one after the other), the step kernel's work
    done inside the Riccati kernel (NEMPC_SOLVER_NO_FUSE_STEP), and the accepted trial's evaluation kept as the next
    iterate's (NEMPC_SOLVER_NO_CARRY: the iterate is evaluated again)."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 20, 300
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    ref = eng.solve(X0, lb=lb, ub=-lb, max_iter=50, return_iterations=True)
    assert int((ref[1] == 0).sum()) > B // 2
    for knob in ("NEMPC_LQ_SPEC", "NEMPC_SOLVER_NO_FUSE_STEP", "NEMPC_SOLVER_NO_CARRY"):
        monkeypatch.setenv(knob, "1")
        got = eng.solve(X0, lb=lb, ub=-lb, max_iter=50, return_iterations=True)
        monkeypatch.delenv(knob)
        assert got[2] == ref[2] and torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3]), knob
        assert torch.equal(got[0], ref[0]), knob
    # inner-loop backtracking: later trials over the still-searching problems only -- same answer with and without compaction
    a = eng.solve(X0, lb=lb, ub=-lb, max_iter=40, linesearch="loop", compact=False)
    b = eng.solve(X0, lb=lb, ub=-lb, max_iter=40, linesearch="loop", compact=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_warm_started_closed_loop_converges_in_fewer_iterations():
    """Closed loop on the network itself (tools/closed_loop_bench.py): each MPC step starts from the previous solution
    shifted by one stage with a small initial barrier parameter; after the first steps every problem converges, in fewer
    iterations than from the reference's cold start."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 20, 64
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    Xh = np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    Zi, cold_it, its = None, None, []
    for k in range(6):
        X = eng.to_device(Xh)
        Z, st, it = eng.solve(X, Zi, lb=lb, ub=-lb, max_iter=80, **({} if Zi is None else {"mu_init": 1e-4}))
        if k == 0:
            cold_it = it
        its.append((it, int((st == 0).sum())))
        z = Z.cpu().numpy()
        xs, us = z[:, :H * nx].reshape(B, H, nx), z[:, H * nx:].reshape(B, H, nu)
        Xh = np.stack([net.forward(np.concatenate([Xh[b], us[b, 0]])) for b in range(B)])   # Discret: x+ = net(x, u)
        xs2 = np.clip(np.concatenate([xs[:, 1:], xs[:, -1:]], axis=1), -2.999, 2.999)
        us2 = np.clip(np.concatenate([us[:, 1:], us[:, -1:]], axis=1), -0.499, 0.499)
        Zi = eng.to_device(np.concatenate([xs2.reshape(B, -1), us2.reshape(B, -1)], axis=1))
    assert all(ok == B for _, ok in its[3:]), its
    assert max(i for i, _ in its[3:]) < cold_it, (its, cold_it)


def test_bounds_are_uploaded_again_when_they_change():
    """The solver keeps the bounds of the last solve on the device and uploads only when they differ: a second solve with
    tighter bounds must respect them, and going back must give the first answer again."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 12, 16
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=2)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    eng.set_objective(Q=np.eye(nx), R=0.05 * np.eye(nu), xref=np.full((H, nx), 1.0))
    X0 = eng.to_device(np.random.default_rng(3).uniform(-0.6, 0.6, size=(B, nx)))
    wide = np.concatenate([np.full(H * nx, -5.0), np.full(H * nu, -1.0)])
    tight = np.concatenate([np.full(H * nx, -0.7), np.full(H * nu, -0.2)])
    Za, sa, _ = eng.solve(X0, lb=wide, ub=-wide, max_iter=80)
    Zb, sb, _ = eng.solve(X0, lb=tight, ub=-tight, max_iter=80)
    Zc, sc, _ = eng.solve(X0, lb=wide, ub=-wide, max_iter=80)
    Zd, sd, _ = eng.solve(X0, lb=wide, ub=-wide, max_iter=80)          # same bounds twice in a row: no upload
    assert torch.equal(Za, Zc) and torch.equal(sa, sc) and torch.equal(Za, Zd)
    zb = Zb.cpu().numpy()
    assert (zb >= tight - 1e-12).all() and (zb <= -tight + 1e-12).all()
    assert np.abs(Za.cpu().numpy()[:, H * nx:]).max() > 0.2 + 1e-3      # the wide solve does use controls beyond the tight limit


def test_solver_refuses_unknown_option_values():
    from pyneuralempc_amd import CallbackEngine
    net = orc.MLP.random(3, [16], 2, seed=1)
    eng = CallbackEngine(net.W, net.b, 5, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4)
    X0 = eng.to_device(np.zeros((4, 2)))
    with pytest.raises(KeyError):
        eng.solve(X0, linesearch="sometimes")
    with pytest.raises(KeyError):
        eng.solve(X0, barrier="dual")
