"""GPU parity at the sizes the performance numbers are quoted on (BASELINE configs[1] dims at B=1024, configs[2],
configs[4]): the FULL launch is checked against the CPU oracle on three 16-problem slices (first, middle, last),
structural zeros must be exact zeros, and the kernel families must agree with each other.  Grid / tile-range /
row -> (problem, step) arithmetic only shows its bugs at these sizes and at ragged batch sizes."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc

pytestmark = pytest.mark.gpu

F64 = dict(rtol=1e-12, atol=1e-12)
DEFAULT = ("f", "grad", "g", "jac_dense")
ALL = ("f", "grad", "g", "jac_dense", "jac_tiles", "jac_sparse")


def _slices(B, k=16):
    k = min(k, B)
    return [slice(s, s + k) for s in sorted({0, max(0, B // 2 - k // 2), B - k})]


def _engine(net, H, nx, nu, B, kernel="auto", integrator="discret", DT=1.0, dtype=torch.float64, box=None):
    from pyneuralempc_amd import CallbackEngine
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integrator, DT=DT, dtype=dtype, device="cuda:0",
                         max_batch=B, kernel=kernel)
    if box is not None:
        eng.set_box_rows(*box)
    return eng


@pytest.mark.parametrize("B", [1024, 256, 1000, 37, 1, 2600])       # 256: configs[1] to the letter
def test_c2_fused_evaluation_full_size(B):
    """configs[1] dims: the one-launch evaluation (rows + dense Jacobian + objective, rows_coopfx_kernel) against the
    oracle and, bit for bit, against the unfused launch sequence of the same handle."""
    H, nx, nu = 20, 2, 1
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = _engine(net, H, nx, nu, B)
    eng.set_objective(Q=[[1.0, 0.2], [0.1, 0.7]], R=[[0.3]], xref=np.linspace(-1, 1, H * nx).reshape(H, nx),
                      uref=0.1, cx=0.05, cu=-0.2, QT=[[2.0, 0.0], [0.3, 1.5]])
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=4)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    fused = {k: v.clone() for k, v in eng.eval(Z, X0, DEFAULT).items()}
    assert eng.last_row_kernel == "rows_coopfx_kernel"
    unf = {k: v.clone() for k, v in eng.eval(Z, X0, ALL).items()}       # sparse requested -> separate assembly launches
    for k in DEFAULT:
        assert torch.equal(fused[k], unf[k]), k
    prob = orc.Problem(net, H, nx, nu, Q=np.array([[1.0, 0.2], [0.1, 0.7]]), R=np.array([[0.3]]),
                       xref=np.linspace(-1, 1, H * nx).reshape(H, nx), uref=np.full((H, nu), 0.1),
                       cx=np.full((H, nx), 0.05), cu=np.full((H, nu), -0.2), QT=np.array([[2.0, 0.0], [0.3, 1.5]]))
    rows, cols = eng.jac_structure()
    mask = np.zeros((eng.m, eng.n), dtype=bool)
    mask[rows, cols] = True
    jac = fused["jac_dense"].cpu().numpy()
    assert np.all(jac[:, ~mask] == 0.0)
    for sl in _slices(B):
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        np.testing.assert_allclose(fused["f"][sl].cpu().numpy(), f, **F64)
        np.testing.assert_allclose(fused["grad"][sl].cpu().numpy(), grad, **F64)
        np.testing.assert_allclose(fused["g"][sl].cpu().numpy(), g, **F64)
        np.testing.assert_allclose(jac[sl], J, **F64)
    # g only / without the objective / with the compact tiles: every output subset of the fused launch
    only = eng.eval(Z, X0, ("g", "jac_dense"))
    assert torch.equal(only["jac_dense"], fused["jac_dense"]) and torch.equal(only["g"], fused["g"])
    wt = eng.eval(Z, X0, ("f", "g", "jac_dense", "jac_tiles"))
    assert torch.equal(wt["jac_tiles"], unf["jac_tiles"]) and torch.equal(wt["f"], fused["f"])
    # the same launch without the dense matrix: objective + compact tiles, and forward passes only (what a line-search
    # trial of the batched solver asks for)
    ct = {k: v.clone() for k, v in eng.eval(Z, X0, ("f", "grad", "g", "jac_tiles")).items()}
    assert eng.last_row_kernel == "rows_coopfx_kernel"
    for k in ("f", "grad", "g"):
        assert torch.equal(ct[k], fused[k]), k
    assert torch.equal(ct["jac_tiles"], unf["jac_tiles"])
    fo = {k: v.clone() for k, v in eng.eval(Z, X0, ("f", "g")).items()}
    assert torch.equal(fo["f"], fused["f"]) and torch.equal(fo["g"], fused["g"])


def test_c2_unity_and_odd_horizon_take_the_right_path():
    """Unity on the fixed-shape kernel; an odd n (H odd, 3 variables per step) cannot be streamed as 16-byte vectors and
    must fall back to the two-launch path with identical results."""
    net = orc.MLP.random(3, [64, 64], 2, seed=1)
    # H = 100 / 150: the objective's table no longer fits the prologue copy (its problems take the kernel's tail path from
    # global memory) and the dense rows go through the LDS row buffer in several chunks per pass
    for H, integ in ((20, "unity"), (7, "discret"), (64, "discret"), (100, "discret"), (150, "unity")):
        B = 130
        eng = _engine(net, H, 2, 1, B, integrator=integ)
        Zh, X0h = orc.synthetic_inputs(B, H, 2, 1, seed=9)
        res = eng.eval_numpy(Zh, X0h)
        prob = orc.Problem(net, H, 2, 1, orc.UNITY if integ == "unity" else orc.DISCRET)
        f, grad, g, J = prob.eval_batch(Zh, X0h)
        np.testing.assert_allclose(res["jac_dense"], J, **F64)
        np.testing.assert_allclose(res["g"], g, **F64)
        np.testing.assert_allclose(res["f"], f, **F64)
        np.testing.assert_allclose(res["grad"], grad, **F64)


def test_c5_full_size_box_rows_and_hessian():
    """configs[4]: B=1024, H=50, box state rows in g / jac (m = 200), fp64, incl. the Lagrangian Hessian callback."""
    B, H, nx, nu = 1024, 50, 2, 1
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    prob = orc.Problem(net, H, nx, nu, box=(-2.0, 2.0))
    out = {}
    for kern in ("mfma", "mfma_tile"):
        eng = _engine(net, H, nx, nu, B, kernel=kern, box=(-2.0, 2.0))
        assert eng.m == 200 and eng.n == 150
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        res = eng.eval(Z, X0, DEFAULT)
        out[kern] = {k: v.cpu().numpy() for k, v in res.items()}
        if kern == "mfma":
            assert eng.last_row_kernel == "rows_coopfx_kernel"
            rows, cols = eng.jac_structure()
            mask = np.zeros((eng.m, eng.n), dtype=bool)
            mask[rows, cols] = True
            assert np.all(out[kern]["jac_dense"][:, ~mask] == 0.0)
            lam = torch.randn(B, eng.m, dtype=torch.float64, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(3))
            sig = torch.full((B,), 0.7, dtype=torch.float64, device="cuda:0")
            hv = eng.hess(Z, X0, lam, sig)["hvals"].cpu().numpy()
            lam_h = lam.cpu().numpy()
            for sl in _slices(B, 4):
                ref = np.stack([prob.hessian_values(Zh[i], X0h[i], lam_h[i], 0.7) for i in range(sl.start, sl.stop)])
                np.testing.assert_allclose(hv[sl], ref, rtol=1e-11, atol=1e-11)
        del eng, res
    for k in DEFAULT:
        np.testing.assert_allclose(out["mfma"][k], out["mfma_tile"][k], **F64)
    for sl in _slices(B):
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        np.testing.assert_allclose(out["mfma"]["f"][sl], f, **F64)
        np.testing.assert_allclose(out["mfma"]["grad"][sl], grad, **F64)
        np.testing.assert_allclose(out["mfma"]["g"][sl], g, **F64)
        np.testing.assert_allclose(out["mfma"]["jac_dense"][sl], J, **F64)


def test_c3_full_size_rk4_fp32():
    """configs[2]: B=1024, H=30, 6 states / 3 controls, MLP 3x128, RK4, fp32 (tolerance 1e-4 relative to the fp64
    oracle): 30,720 rows on the cooperative matrix-core kernel and on the wave-per-tile one."""
    B, H, nx, nu, DT = 1024, 30, 6, 3, 0.1
    net = orc.MLP.random(9, [128, 128, 128], 6, seed=0)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    prob = orc.Problem(net, H, nx, nu, orc.RK4, DT)
    out = {}
    for kern in ("mfma", "mfma_tile"):
        eng = _engine(net, H, nx, nu, B, kernel=kern, integrator="rk4", DT=DT, dtype=torch.float32)
        res = eng.eval(eng.to_device(Zh), eng.to_device(X0h), DEFAULT)
        out[kern] = {k: v.to("cpu", torch.float64).numpy() for k, v in res.items()}
        if kern == "mfma":
            assert eng.last_row_kernel.startswith("rows_coop_kernel")
            rows, cols = eng.jac_structure()
            mask = np.zeros((eng.m, eng.n), dtype=bool)
            mask[rows, cols] = True
            assert np.all(out[kern]["jac_dense"][:, ~mask] == 0.0)
        del eng, res

    def close(a, b, what):
        err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
        assert err < 1e-4, f"{what}: max rel err {err:.2e}"

    for k in DEFAULT:
        close(out["mfma"][k], out["mfma_tile"][k], f"coop vs wave-tile {k}")
    for sl in _slices(B):
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        for kern in out:
            close(out[kern]["f"][sl], f, "f")
            close(out[kern]["grad"][sl], grad, "grad")
            close(out[kern]["g"][sl], g, "g")
            close(out[kern]["jac_dense"][sl], J, "jac")


@pytest.mark.parametrize("box", [None, (-1.5, 1.5)])
def test_fused_launch_sweep_of_small_shapes(box):
    """The one-launch evaluation over horizons and batch sizes around its internal boundaries (one vector per row side,
    rows that straddle problems inside a tile, a single tile, several passes per workgroup), with and without box rows,
    every output subset that takes the fused kernel -- against the oracle."""
    net = orc.MLP.random(3, [64, 64], 2, seed=3)
    for H in (2, 4, 6, 10, 16, 32):
        for B in (1, 3, 17, 700):
            eng = _engine(net, H, 2, 1, B, box=box)
            eng.set_objective(Q=[[1.0, 0.1], [0.1, 0.5]], R=[[0.2]], cx=0.03, cu=-0.1)
            Zh, X0h = orc.synthetic_inputs(B, H, 2, 1, seed=H + B)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            prob = orc.Problem(net, H, 2, 1, Q=np.array([[1.0, 0.1], [0.1, 0.5]]), R=np.array([[0.2]]),
                               cx=np.full((H, 2), 0.03), cu=np.full((H, 1), -0.1), box=box)
            f, grad, g, J = prob.eval_batch(Zh, X0h)
            res = {k: v.cpu().numpy() for k, v in eng.eval(Z, X0, DEFAULT).items()}
            assert eng.last_row_kernel == "rows_coopfx_kernel", (H, B)
            np.testing.assert_allclose(res["f"], f, **F64)
            np.testing.assert_allclose(res["grad"], grad, **F64)
            np.testing.assert_allclose(res["g"], g, **F64)
            np.testing.assert_allclose(res["jac_dense"], J, **F64)
            sub = {k: v.cpu().numpy() for k, v in eng.eval(Z, X0, ("f", "g")).items()}
            np.testing.assert_allclose(sub["f"], f, **F64)
            np.testing.assert_allclose(sub["g"], g, **F64)
            ct = {k: v.cpu().numpy() for k, v in eng.eval(Z, X0, ("grad", "g", "jac_tiles")).items()}
            np.testing.assert_allclose(ct["grad"], grad, **F64)
            # the compact tiles against the dense matrix they are the blocks of
            T = ct["jac_tiles"]                       # (B, H, nx, nx + nu)
            for t in range(H):
                np.testing.assert_allclose(T[:, t, :, 2:], J[:, 2 * t:2 * t + 2, 2 * H + t:2 * H + t + 1], **F64)
                if t >= 1:
                    np.testing.assert_allclose(T[:, t, :, :2], J[:, 2 * t:2 * t + 2, 2 * (t - 1):2 * t], **F64)
            del eng


@pytest.mark.parametrize("dims", [(20, 9001, None), (50, 3301, (-2.0, 2.0))])
def test_fused_evaluation_at_batches_of_many_passes_per_workgroup(dims):
    """Batches beyond 8192 row tiles (every workgroup of the one-launch kernel runs many passes; ragged last tile): the
    fixed-shape kernel keeps the launch, agrees with the wave-per-tile family to rounding and with the oracle on the
    first / middle / last slices, structural zeros exact."""
    H, B, box = dims
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = _engine(net, H, 2, 1, B, box=box)
    ref = _engine(net, H, 2, 1, B, kernel="mfma_tile", box=box)
    Zh, X0h = orc.synthetic_inputs(B, H, 2, 1, seed=11)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    a = eng.eval(Z, X0, DEFAULT)
    assert eng.last_row_kernel == "rows_coopfx_kernel" and (B * H + 15) // 16 > 8192
    b = ref.eval(Z, X0, DEFAULT)
    for k in DEFAULT:
        assert float((a[k] - b[k]).abs().max()) < 1e-12, k
    prob = orc.Problem(net, H, 2, 1, box=box)
    for sl in _slices(B):
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        np.testing.assert_allclose(a["f"][sl].cpu().numpy(), f, **F64)
        np.testing.assert_allclose(a["grad"][sl].cpu().numpy(), grad, **F64)
        np.testing.assert_allclose(a["g"][sl].cpu().numpy(), g, **F64)
        Jd = a["jac_dense"][sl].cpu().numpy()
        np.testing.assert_allclose(Jd, J, **F64)
        assert np.array_equal(Jd == 0.0, J == 0.0)


@pytest.mark.parametrize("num_cus", [32, 64, 304])
def test_launch_geometry_follows_the_device_cu_count(num_cus, monkeypatch):
    """The chip-filling launches are sized from the device's CU count (hipDeviceProp_t::multiProcessorCount, or
    NEMPC_NUM_CUS for this test), not from a literal 256: with a different count every kernel family still covers every
    tile and -- per-problem arithmetic does not depend on which workgroup owns a tile -- gives the same bits."""
    from pyneuralempc_amd import _lib
    B, H, nx, nu = 520, 20, 2, 1
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=4)
    lamh = np.random.default_rng(5).normal(size=(B, H * nx))
    ref = {}
    for cus in (None, num_cus):
        if cus is None:
            monkeypatch.delenv("NEMPC_NUM_CUS", raising=False)
        else:
            monkeypatch.setenv("NEMPC_NUM_CUS", str(cus))
        for kernel in ("auto", "mfma_tile"):
            for integ, dt in (("discret", torch.float64), ("rk4", torch.float32)):
                eng = _engine(net, H, nx, nu, B, kernel=kernel, integrator=integ, DT=0.1, dtype=dt)
                want_cus = cus if cus is not None else torch.cuda.get_device_properties(0).multi_processor_count
                assert _lib.load().nempc_num_cus(eng._handle) == want_cus
                Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
                out = {k: v.clone() for k, v in eng.eval(Z, X0, ALL).items()}
                fused = {k: v.clone() for k, v in eng.eval(Z, X0, DEFAULT).items()}
                hv = eng.hess(Z, X0, eng.to_device(lamh), torch.ones(B, dtype=dt, device="cuda:0"))["hvals"].clone()
                key = (kernel, integ)
                if cus is None:
                    ref[key] = (out, fused, hv)
                else:
                    for k in ALL:
                        assert torch.equal(out[k], ref[key][0][k]), (key, k)
                    for k in DEFAULT:
                        assert torch.equal(fused[k], ref[key][1][k]), (key, k)
                    assert torch.equal(hv, ref[key][2]), key


@pytest.mark.parametrize("dtype,hidden", [(torch.float64, [30, 30]), (torch.float32, [30, 30]), (torch.float32, [64, 64])])
def test_compiled_shape_table_beyond_the_baseline_shape(dtype, hidden):
    """The table of compiled fixed shapes (csrc/kernels_mfma_typed.inc, FxTable): the reference's own example network
    3 -> 30 -> 30 -> 2 (examples/lotka_volterra/run.py:64-98, nn_model.h5) in both precisions and 2/1 2 x 64 in fp32 take
    rows_coopfx_kernel / rowhess_coopfx_kernel -- every output subset of the one-launch evaluation (dense, sparse, tiles,
    forward only), box rows, ragged batches, the Hessian callbacks and the batched solver -- against the oracle."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu = 2, 1
    net = orc.MLP.random(nx + nu, hidden, nx, seed=11)
    f64 = dtype == torch.float64
    tol = dict(rtol=1e-12, atol=1e-12) if f64 else dict(rtol=2e-4, atol=2e-4)
    # (fp32: the dense rows leave as 16-byte vectors of four, so the one-launch dense evaluation needs n = 3 H divisible by 4)
    for H, B, box in ((10 if f64 else 8, 1, None), (10 if f64 else 12, 37, None), (20, 300, (-1.5, 1.5)), (12, 1024, None)):
        prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=np.array([[1.0, 0.1], [0.1, 0.5]]), R=np.array([[0.2]]),
                           cx=np.full((H, nx), 0.03), cu=np.full((H, nu), -0.1), box=box)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="discret", dtype=dtype, device="cuda:0", max_batch=B)
        eng.set_objective(Q=[[1.0, 0.1], [0.1, 0.5]], R=[[0.2]], cx=0.03, cu=-0.1)
        if box:
            eng.set_box_rows(*box)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=H + B)
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        sl = slice(0, min(B, 24))
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        res = {k: v.to("cpu", torch.float64).numpy() for k, v in eng.eval(Z, X0, DEFAULT).items()}
        assert eng.last_row_kernel == "rows_coopfx_kernel", (H, B)
        for k, ref in (("f", f), ("grad", grad), ("g", g), ("jac_dense", J)):
            np.testing.assert_allclose(res[k][sl], ref, **tol, err_msg=f"{k} H={H} B={B}")
        rows, cols = eng.jac_structure()
        mask = np.zeros((eng.m, eng.n), dtype=bool)
        mask[rows, cols] = True
        assert np.all(res["jac_dense"][:, ~mask] == 0.0)
        sp = {k: v.to("cpu", torch.float64).numpy() for k, v in eng.eval(Z, X0, ("f", "grad", "g", "jac_sparse")).items()}
        assert eng.last_row_kernel == "rows_coopfx_kernel+sparse"
        assert np.array_equal(sp["jac_sparse"], res["jac_dense"][:, rows, cols]) and np.array_equal(sp["g"], res["g"])
        ct = {k: v.to("cpu", torch.float64).numpy() for k, v in eng.eval(Z, X0, ("f", "g", "jac_tiles")).items()}
        assert eng.last_row_kernel == "rows_coopfx_kernel" and np.array_equal(ct["f"], res["f"])
        for i in range(min(B, 3)):
            _, A, Bt = prob.tiles_AB(Zh[i], X0h[i])
            np.testing.assert_allclose(ct["jac_tiles"][i][:, :, :nx], A, **tol)
            np.testing.assert_allclose(ct["jac_tiles"][i][:, :, nx:], Bt, **tol)
        fo = eng.eval(Z, X0, ("f", "g"))
        assert np.array_equal(fo["g"].to("cpu", torch.float64).numpy(), res["g"])
        # Hessian callbacks
        rng = np.random.default_rng(1)
        lam, sig = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
        hv = eng.hess(Z, X0, eng.to_device(lam), eng.to_device(sig))["hvals"].to("cpu", torch.float64).numpy()
        assert eng.last_hess_kernel == "rowhess_coopfx_kernel"
        wgt = rng.uniform(0.2, 1.5, size=(B, H * nx))
        gv = eng.hess_gn(Z, X0, eng.to_device(wgt), eng.to_device(sig))["hvals"].to("cpu", torch.float64).numpy()
        for i in range(min(B, 6)):
            ref = prob.hessian_values(Zh[i], X0h[i], lam[i], sig[i])
            np.testing.assert_allclose(hv[i], ref, rtol=0, atol=(1e-10 if f64 else 2e-3) * max(1.0, np.abs(ref).max()))
            refg = prob.gauss_newton_values(Zh[i], X0h[i], wgt[i], sig[i])
            np.testing.assert_allclose(gv[i], refg, rtol=0, atol=(1e-10 if f64 else 2e-3) * max(1.0, np.abs(refg).max()))
    # the batched solver on this shape (its trial-point launch is the fixed-shape Hessian kernel's EV variant)
    H, B = 10, 64
    net2 = orc.MLP.random(nx + nu, hidden, nx, seed=11)
    net2.W[-1] *= 0.2
    net2.b[-1] *= 0.2
    prob = orc.Problem(net2, H, nx, nu, orc.DISCRET, Q=np.eye(nx), R=0.1 * np.eye(nu))
    eng = CallbackEngine(net2.W, net2.b, H, nx, nu, integrator="discret", dtype=dtype, device="cuda:0", max_batch=B)
    eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
    X0 = np.random.default_rng(2).uniform(-0.5, 0.5, size=(B, nx))
    lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
    Zs, status, iters = eng.solve(eng.to_device(X0), lb=lb, ub=-lb, max_iter=200)
    Zs, status = Zs.to("cpu", torch.float64).numpy(), status.cpu().numpy()
    assert (status == 0).mean() >= 0.9
    for i in np.nonzero(status == 0)[0][:16]:
        assert np.abs(prob.constraints(Zs[i], X0[i])).max() < (1e-6 if f64 else 2e-3)


@pytest.mark.parametrize("case", [("discret", [256, 256], 2, 1, 20, 1024, torch.float64), ("rk4", [192, 160, 128], 6, 3, 30, 1024, torch.float32),
                                  ("unity", [320], 3, 2, 12, 2600, torch.float64)])
def test_layered_path_full_size(case):
    """The layer-at-a-time GEMM path at the batch sizes its numbers are quoted on: rows and Lagrangian blocks against the
    oracle on three slices of the batch, exact structural zeros, and -- the size-independent property -- a permutation of
    the batch permutes the results bit for bit (rows of different problems share GEMM blocks, chunk boundaries and
    feature-block partial sums; none of that may leak from one problem into another)."""
    integ, hidden, nx, nu, H, B, dt = case
    f64 = dt == torch.float64
    DT = 0.1 if integ == "rk4" else 1.0
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[integ]
    net = orc.MLP.random(nx + nu, hidden, nx, seed=5)
    prob = orc.Problem(net, H, nx, nu, kind, DT)
    eng = _engine(net, H, nx, nu, B, "layered", integ, DT, dt)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=2)
    lamh = np.random.default_rng(3).normal(size=(B, eng.m))
    one = eng.to_device(np.ones(B))

    def run(Zx, X0x, lx):
        res = eng.eval_numpy(Zx, X0x, want=("g", "jac_dense", "jac_tiles"))
        hv = eng.hess(eng.to_device(Zx), eng.to_device(X0x), eng.to_device(lx), one)["hvals"].cpu().numpy()
        return res, hv
    res, hv = run(Zh, X0h, lamh)
    assert eng.last_row_kernel == "layered_gemm_kernel" and eng.last_hess_kernel.endswith("layered_gemm_kernel")
    tol = dict(rtol=1e-10, atol=1e-10) if f64 else dict(rtol=2e-3, atol=2e-3)
    for sl in _slices(B, 4 if integ == "rk4" else 8):
        f, grad, g, J = prob.eval_batch(Zh[sl], X0h[sl])
        np.testing.assert_allclose(res["g"][sl], g, **tol)
        np.testing.assert_allclose(res["jac_dense"][sl], J, **tol)
        assert np.array_equal(res["jac_dense"][sl] != 0, J != 0)
        i = sl.start
        ref = prob.hessian_values(Zh[i], X0h[i], lamh[i], 1.0)
        np.testing.assert_allclose(hv[i], ref, rtol=0, atol=(1e-9 if f64 else 1e-2) * max(1.0, np.abs(ref).max()))
    perm = np.random.default_rng(4).permutation(B)
    res_p, hv_p = run(Zh[perm], X0h[perm], lamh[perm])
    assert np.array_equal(res_p["g"], res["g"][perm]) and np.array_equal(res_p["jac_tiles"], res["jac_tiles"][perm])
    assert np.array_equal(hv_p, hv[perm])
