"""GPU parity: the HIP path (through the C ABI) against the reference-generated golden vectors and
against the CPU oracle on seeded inputs.  fp64 tolerance 1e-12 abs + 1e-12 rel;
fp32 configs 1e-4 relative to the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc
from helpers import ACT_MIXED_NAMES, ACT_RUNTIME_NAMES, ACT_UNIFORM_NAMES, CASE_NAMES, WIDE_DEEP_NAMES, ZBASED_NAMES, case_activations, case_extra, load_case, oracle_problem

pytestmark = pytest.mark.gpu

F64 = dict(rtol=1e-12, atol=1e-12)
KIND_NAME = {0: "discret", 1: "unity", 2: "rk4"}


def _engine(d, W, b, dtype, kernel, max_batch=8):
    from pyneuralempc_amd import CallbackEngine
    ex = case_extra(d)
    eng = CallbackEngine(W, b, int(d["H"]), int(d["nx"]), int(d["nu"]), integrator=KIND_NAME[int(d["kind"])],
                         DT=float(d["DT"]), dtype=dtype, device="cuda:0", max_batch=max_batch, kernel=kernel,
                         n_extra=0 if ex is None else ex.shape[1], activations=case_activations(d))
    if ex is not None:   # the same parameters for every problem of the golden batch
        eng.bind_extra(eng.to_device(np.broadcast_to(ex[None], (max_batch,) + ex.shape).copy()))
    eng.set_objective(Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"])
    if int(d["has_box"]):
        eng.set_box_rows(d["box_lo"], d["box_hi"])
    return eng


def _f32_close(a, b, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1.0, np.abs(b).max())
    err = np.abs(a - b).max() / scale
    assert err < 1e-4, f"{what}: max rel err {err:.3e}"


ALL = ("f", "grad", "g", "jac_dense", "jac_tiles", "jac_sparse")


# every golden case x every kernel family; per-layer activation mixes run on the generic kernel only
_FP64_CASES = ([(n, k) for n in CASE_NAMES + ACT_UNIFORM_NAMES for k in ("valu", "mfma", "mfma_tile", "layered")] +
               [(n, k) for n in ACT_MIXED_NAMES + WIDE_DEEP_NAMES for k in ("valu", "layered")] +
               [(n, k) for n in ACT_RUNTIME_NAMES for k in ("mfma", "mfma_tile")] +
               [(n, k) for n in ACT_RUNTIME_NAMES if n not in ACT_MIXED_NAMES + WIDE_DEEP_NAMES for k in ("valu", "layered")] +
               [(n, k) for n in ZBASED_NAMES for k in ("layered", "valu")])        # (swish / gelu ...: derivatives from the pre-activation)


@pytest.mark.parametrize("name,kernel", _FP64_CASES)
def test_golden_fp64(name, kernel):
    d, W, b = load_case(name)
    eng = _engine(d, W, b, torch.float64, kernel)
    assert eng.kernel_variant == kernel
    res = eng.eval_numpy(d["Z"], d["X0"], want=ALL)
    np.testing.assert_allclose(res["f"], d["f"], **F64)
    np.testing.assert_allclose(res["grad"], d["grad"], **F64)
    np.testing.assert_allclose(res["g"], d["g"], **F64)
    np.testing.assert_allclose(res["jac_dense"], d["jac"], **F64)
    # structural zeros are exact zeros, and the sparse / tile contracts agree with the dense one bit for bit
    assert np.array_equal(res["jac_dense"] != 0, d["jac"] != 0)
    rows, cols = eng.jac_structure()
    assert np.array_equal(res["jac_sparse"], res["jac_dense"][:, rows, cols])
    prob = oracle_problem(d, W, b)
    for i in range(d["Z"].shape[0]):
        _, A, Bt = prob.tiles_AB(d["Z"][i], d["X0"][i])
        np.testing.assert_allclose(res["jac_tiles"][i][:, :, :prob.nx], A, **F64)
        np.testing.assert_allclose(res["jac_tiles"][i][:, :, prob.nx:], Bt, **F64)
    cl, cu = eng.constraint_bounds()
    np.testing.assert_array_equal(cl, d["cl"])
    np.testing.assert_array_equal(cu, d["cu_bound"])


@pytest.mark.parametrize("name,kernel", [(n, k) for n in ["c2_discret", "c3_rk4", "c3_discret", "c5_box", "odd_dims"] +
                                         ACT_UNIFORM_NAMES for k in ("valu", "mfma", "mfma_tile", "layered")] +
                         [(n, k) for n in ACT_MIXED_NAMES + WIDE_DEEP_NAMES for k in ("valu", "layered")] +
                         [(n, k) for n in ACT_RUNTIME_NAMES for k in ("mfma", "mfma_tile")] +
                         [(n, k) for n in ZBASED_NAMES for k in ("layered", "valu")])
def test_golden_fp32(name, kernel):
    d, W, b = load_case(name)
    eng = _engine(d, W, b, torch.float32, kernel)
    res = eng.eval_numpy(d["Z"], d["X0"], want=ALL)
    for k, ref in (("f", d["f"]), ("grad", d["grad"]), ("g", d["g"]), ("jac_dense", d["jac"])):
        _f32_close(res[k], ref, f"{name}/{k}")
    assert np.array_equal(res["jac_dense"] != 0, d["jac"] != 0)


@pytest.mark.parametrize("name,kernel",
                         [(n, k) for n in [c for c in CASE_NAMES if c not in ("c3_rk4", "odd_dims", "c3_discret")] +
                          [c for c in ACT_UNIFORM_NAMES if c.endswith("_c2")] for k in ("valu", "mfma", "mfma_tile", "layered")] +
                         [(n, k) for n in ACT_MIXED_NAMES + WIDE_DEEP_NAMES for k in ("valu", "layered")] +
                         [(n, k) for n in ACT_RUNTIME_NAMES if n != "act_mix_c3_rk4" for k in ("mfma", "mfma_tile")] +
                         [(n, k) for n in ACT_RUNTIME_NAMES if n not in ACT_MIXED_NAMES + WIDE_DEEP_NAMES + ["act_mix_c3_rk4"]
                          for k in ("valu", "layered")] +
                         [(n, k) for n in ZBASED_NAMES for k in ("layered", "valu")])
def test_golden_hessian_fp64(name, kernel):
    d, W, b = load_case(name)
    eng = _engine(d, W, b, torch.float64, kernel)
    Z, X0 = eng.to_device(d["Z"]), eng.to_device(d["X0"])
    lam, sig = eng.to_device(d["lam"]), eng.to_device(d["sigma"])
    out = eng.hess(Z, X0, lam, sig, want=("hvals", "hdense"))
    hd = out["hdense"].cpu().numpy()
    hv = out["hvals"].cpu().numpy()
    np.testing.assert_allclose(hd, d["hdense"], rtol=1e-11, atol=1e-12)
    rows, cols = eng.hess_structure()
    assert np.array_equal(hv, hd[:, rows, cols])
    # the reference's sampled pattern is inside ours and its values agree
    ours = set(zip(rows.tolist(), cols.tolist()))
    assert set(zip(d["h_rows"].tolist(), d["h_cols"].tolist())) <= ours
    np.testing.assert_allclose(hd[:, d["h_rows"], d["h_cols"]], d["hvals"], rtol=1e-11, atol=1e-12)
    prob = oracle_problem(d, W, b)
    orows, ocols = prob.hessian_structure()
    assert np.array_equal(rows, orows) and np.array_equal(cols, ocols)


@pytest.mark.parametrize("cfg", [(6, 3, [128, 128, 128], 30, 7), (3, 2, [48, 32], 7, 5), (12, 4, [96, 96], 4, 3),
                                 (2, 1, [64, 64], 50, 40), (12, 9, [64, 64], 3, 6), (16, 16, [32], 2, 3)])
def test_hessian_against_oracle_seeded(cfg):
    """Lagrangian-Hessian blocks of the matrix-core kernel vs the oracle (which carries no golden for these
    shapes: the reference's own integrator Hessian only exists for nx+nu = 3) and vs the generic kernel."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, B = cfg
    net = orc.MLP.random(nx + nu, hidden, nx, seed=3)
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    lamh = np.random.default_rng(6).normal(size=(B, prob.m))
    sigh = np.random.default_rng(7).uniform(0.0, 2.0, size=B)
    dense = {}
    for kernel in ("mfma", "valu"):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=kernel)
        out = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lamh), eng.to_device(sigh),
                       want=("hvals", "hdense"))
        dense[kernel] = out["hdense"].cpu().numpy()
        assert np.array_equal(dense[kernel], np.transpose(dense[kernel], (0, 2, 1)))   # exactly symmetric
    np.testing.assert_allclose(dense["mfma"], dense["valu"], rtol=1e-10, atol=1e-11)
    for i in range(min(B, 3)):
        np.testing.assert_allclose(dense["mfma"][i], prob.lagrangian_hessian(Zh[i], X0h[i], lamh[i], sigh[i]),
                                   rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("cfg", [(6, 3, [128, 128, 128], 30, 0.1, 2), (3, 2, [48, 32], 7, 0.05, 5), (2, 1, [64, 64], 20, 0.5, 9),
                                 (1, 1, [20], 3, 0.2, 4), (12, 4, [96, 96], 4, 0.2, 3)])
def test_rk4_hessian_against_oracle(cfg):
    """RK4 Lagrangian Hessian for general dims (the reference's own RK4Integrator.hessian is hard-wired to
    nx+nu = 3, rk4.py:246; the golden c2_rk4 case above pins that one, the oracle the rest).  Matrix-core pipeline
    (stage records -> stage multipliers -> contracted stage Hessians -> congruence sum, csrc/kernels_rk4hess.hip)
    and the generic kernel."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, DT, B = cfg
    net = orc.MLP.random(nx + nu, hidden, nx, seed=3)
    prob = orc.Problem(net, H, nx, nu, orc.RK4, DT)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    lamh = np.random.default_rng(6).normal(size=(B, prob.m))
    sigh = np.random.default_rng(7).uniform(0.0, 2.0, size=B)
    dense = {}
    for kernel in ("mfma", "valu"):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=DT, dtype=torch.float64, device="cuda:0",
                             max_batch=B, kernel=kernel)
        out = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lamh), eng.to_device(sigh),
                       want=("hvals", "hdense"))
        hd = dense[kernel] = out["hdense"].cpu().numpy()
        assert np.array_equal(hd, np.transpose(hd, (0, 2, 1)))
        for i in range(B):
            np.testing.assert_allclose(hd[i], prob.lagrangian_hessian(Zh[i], X0h[i], lamh[i], sigh[i]),
                                       rtol=1e-10, atol=1e-11)
        # the evaluation callbacks are untouched by the Hessian pipeline's scratch use
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        np.testing.assert_allclose(res["jac_dense"][0], prob.jacobian(Zh[0], X0h[0]), **F64)
    np.testing.assert_allclose(dense["mfma"], dense["valu"], rtol=1e-10, atol=1e-11)


def test_rk4_hessian_fp32_c3_dims():
    from pyneuralempc_amd import CallbackEngine
    nx, nu, hidden, H, DT, B = 6, 3, [128, 128, 128], 30, 0.1, 5
    net = orc.MLP.random(nx + nu, hidden, nx, seed=3)
    prob = orc.Problem(net, H, nx, nu, orc.RK4, DT)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    lamh = np.random.default_rng(6).normal(size=(B, prob.m))
    sigh = np.ones(B)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=DT, dtype=torch.float32, device="cuda:0",
                         max_batch=B, kernel="mfma")
    hd = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lamh), eng.to_device(sigh),
                  want=("hdense",))["hdense"].cpu().numpy().astype(np.float64)
    for i in (0, B - 1):
        ref = prob.lagrangian_hessian(Zh[i], X0h[i], lamh[i], 1.0)
        assert np.abs(hd[i] - ref).max() / max(1.0, np.abs(ref).max()) < 1e-4


@pytest.mark.parametrize("kernel", ["valu", "mfma", "mfma_tile"])
@pytest.mark.parametrize("cfg", [
    # (nx, nu, hidden, H, kind, DT, box, B)   B*H deliberately not a multiple of 16
    (2, 1, [64, 64], 20, orc.DISCRET, 1.0, None, 37),
    (2, 1, [64, 64], 50, orc.DISCRET, 1.0, (-2.0, 2.0), 9),
    (6, 3, [128, 128, 128], 30, orc.RK4, 0.1, None, 5),
    (4, 2, [32], 3, orc.UNITY, 1.0, None, 11),
    (3, 2, [48, 32], 7, orc.RK4, 0.05, None, 13),
    (1, 1, [20, 20, 20], 5, orc.DISCRET, 1.0, None, 7),
    (16, 1, [64], 2, orc.DISCRET, 1.0, None, 3),     # nx at the MFMA-path limit, nin = 17: two input blocks (tile kernel)
    (10, 9, [64, 64], 3, orc.RK4, 0.1, None, 4),      # nin = 19, RK4 chain on two input blocks
    (16, 16, [32], 2, orc.UNITY, 1.0, None, 2),       # nin = 32: widest matrix-core shape
    (12, 4, [96, 96], 4, orc.RK4, 0.2, None, 3),      # nin = 16: MFMA-path limit
])
def test_against_oracle_seeded_fp64(cfg, kernel):
    nx, nu, hidden, H, kind, DT, box, B = cfg
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd._lib import NempcError
    net = orc.MLP.random(nx + nu, hidden, nx, seed=3)
    prob = orc.Problem(net, H, nx, nu, kind, DT, box=box)
    try:
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=KIND_NAME[kind], DT=DT, dtype=torch.float64,
                             device="cuda:0", max_batch=B, kernel=kernel)
    except NempcError:
        assert kernel.startswith("mfma") and nx + nu > 32   # documented shape gate of the matrix-core kernels
        return
    if box is not None:
        eng.set_box_rows(*box)
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    res = eng.eval_numpy(Z, X0, want=ALL)
    f, grad, g, jac = prob.eval_batch(Z, X0)
    np.testing.assert_allclose(res["f"], f, **F64)
    np.testing.assert_allclose(res["grad"], grad, **F64)
    np.testing.assert_allclose(res["g"], g, rtol=1e-12, atol=2e-12)
    np.testing.assert_allclose(res["jac_dense"], jac, rtol=1e-11, atol=2e-12)


def test_batch_invariance_and_reuse():
    """A problem's result does not depend on its position in the batch nor on the batch size
    (bit-exact), and repeated calls on reused output buffers are deterministic."""
    d, W, b = load_case("c2_discret")
    eng = _engine(d, W, b, torch.float64, "mfma", max_batch=64)
    Z, X0 = orc.synthetic_inputs(50, 20, 2, 1, seed=8)
    full = {k: v.copy() for k, v in eng.eval_numpy(Z, X0, want=ALL).items()}
    perm = np.random.default_rng(0).permutation(50)
    shuf = eng.eval_numpy(Z[perm], X0[perm], want=ALL)
    for k in ALL:
        assert np.array_equal(shuf[k], full[k][perm]), k
    one = eng.eval_numpy(Z[17:18], X0[17:18], want=ALL)
    for k in ALL:
        assert np.array_equal(one[k][0], full[k][17]), k
    again = eng.eval_numpy(Z, X0, want=ALL)
    for k in ALL:
        assert np.array_equal(again[k], full[k]), k


def test_full_size_properties_c2():
    """BASELINE sizes (B=1024, H=20): directional finite differences of g and f against the device
    Jacobian / gradient, valu-vs-mfma agreement, exact structural zeros."""
    B, H, nx, nu = 1024, 20, 2, 1
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    from pyneuralempc_amd import CallbackEngine
    engs = {k: CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B, kernel=k)
            for k in ("valu", "mfma", "mfma_tile")}
    Z, X0 = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    r = {k: {kk: v.copy() for kk, v in e.eval_numpy(Z, X0, want=ALL).items()} for k, e in engs.items()}
    for k in ("f", "grad", "g", "jac_dense"):
        np.testing.assert_allclose(r["mfma"][k], r["valu"][k], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(r["mfma_tile"][k], r["valu"][k], rtol=1e-12, atol=1e-12)
    rows, cols = engs["mfma"].jac_structure()
    mask = np.zeros((40, 60), dtype=bool)
    mask[rows, cols] = True
    assert np.all(r["mfma"]["jac_dense"][:, ~mask] == 0.0)
    v = np.random.default_rng(2).normal(size=Z.shape)
    eps = 1e-6
    gp = engs["mfma"].eval_numpy(Z + eps * v, X0, want=("f", "g"))
    gp = {k: x.copy() for k, x in gp.items()}
    gm = engs["mfma"].eval_numpy(Z - eps * v, X0, want=("f", "g"))
    jv = np.einsum("bmn,bn->bm", r["mfma"]["jac_dense"], v)
    assert np.abs((gp["g"] - gm["g"]) / (2 * eps) - jv).max() < 1e-7
    gv = np.einsum("bn,bn->b", r["mfma"]["grad"], v)
    assert np.abs((gp["f"] - gm["f"]) / (2 * eps) - gv).max() < 1e-6
    # spot-check a slice against the oracle
    prob = orc.Problem(net, H, nx, nu)
    f, grad, g, jac = prob.eval_batch(Z[500:516], X0[500:516])
    np.testing.assert_allclose(r["mfma"]["jac_dense"][500:516], jac, **F64)
    np.testing.assert_allclose(r["mfma"]["g"][500:516], g, **F64)


def test_edge_cases():
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd._lib import NempcError
    net = orc.MLP.random(3, [16], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, 4, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=2)
    # empty batch
    Z0 = torch.empty(0, eng.n, dtype=torch.float64, device="cuda:0")
    X00 = torch.empty(0, 2, dtype=torch.float64, device="cuda:0")
    assert eng.eval(Z0, X00)["g"].shape == (0, eng.m)
    # wrong dtype / shape are refused on the host
    with pytest.raises(ValueError):
        eng.eval(torch.zeros(1, eng.n, dtype=torch.float32, device="cuda:0"), torch.zeros(1, 2, dtype=torch.float64, device="cuda:0"))
    with pytest.raises(ValueError):
        eng.eval(torch.zeros(1, eng.n + 1, dtype=torch.float64, device="cuda:0"), torch.zeros(1, 2, dtype=torch.float64, device="cuda:0"))
    # growth beyond max_batch re-creates the handle transparently
    Z, X0 = orc.synthetic_inputs(33, 4, 2, 1, seed=2)
    res = eng.eval_numpy(Z, X0)
    prob = orc.Problem(net, 4, 2, 1)
    np.testing.assert_allclose(res["jac_dense"], prob.eval_batch(Z, X0)[3], **F64)
    # bad construction arguments come back as library errors / ValueError, not crashes
    with pytest.raises(ValueError):
        CallbackEngine(net.W, net.b, 4, 3, 1, device="cuda:0")
    with pytest.raises(NempcError):
        CallbackEngine(net.W, net.b, 0, 2, 1, device="cuda:0")


def test_bound_launcher_tracks_input_contents():
    """engine.bind: the pre-validated launcher re-reads Z / X0 each call and writes the same output tensors."""
    d, W, b = load_case("c2_discret")
    eng = _engine(d, W, b, torch.float64, "auto")
    Z, X0 = eng.to_device(d["Z"]), eng.to_device(d["X0"])
    launch, outs = eng.bind(Z, X0)
    launch()
    torch.cuda.synchronize()
    np.testing.assert_allclose(outs["jac_dense"].cpu().numpy(), d["jac"], **F64)
    Z2, X02 = orc.synthetic_inputs(Z.shape[0], 20, 2, 1, seed=31)
    Z.copy_(eng.to_device(Z2)); X0.copy_(eng.to_device(X02))
    launch()
    torch.cuda.synchronize()
    prob = oracle_problem(d, W, b)
    f, grad, g, jac = prob.eval_batch(Z2, X02)
    np.testing.assert_allclose(outs["jac_dense"].cpu().numpy(), jac, **F64)
    np.testing.assert_allclose(outs["f"].cpu().numpy(), f, **F64)


@pytest.mark.parametrize("name", ["c2_discret", "c3_rk4", "c5_box", "odd_dims", "tvp_p_rk4"])
def test_defect_only_evaluation_matches_full_evaluation(name):
    """constraints() on its own: the matrix-core kernel runs forward only (no reverse sweeps, no tiles); the
    defects must be the ones of the full evaluation, bit for bit on the same kernel family."""
    d, W, b = load_case(name)
    for kernel in ("mfma", "mfma_tile", "valu"):
        eng = _engine(d, W, b, torch.float64, kernel)
        only_g = eng.eval_numpy(d["Z"], d["X0"], want=("g",))["g"]
        np.testing.assert_allclose(only_g, d["g"], **F64)
        if kernel == "mfma_tile":
            full = eng.eval_numpy(d["Z"], d["X0"], want=("g", "jac_tiles"))["g"]
            assert np.array_equal(only_g, full)


@pytest.mark.parametrize("name", ["c1_discret", "c2_discret", "c2_unity", "c2_rk4", "c5_box", "odd_dims", "c3_rk4", "c3_discret", "h1",
                                  "act_relu_c2", "act_elu_c3"])
def test_sparse_contract_fused_launch(name):
    """f, grad, g and the band-pattern Jacobian values without the dense matrix: ONE launch -- the row kernel writes the
    values in nempc_jac_structure order itself (fixed-shape kernel on the compiled shape, cooperative kernel elsewhere); no
    tile round trip, no assembly launch (reference: dense (m, n) to cyipopt, optimizer/ipopt.py:88-96)."""
    d, W, b = load_case(name)
    for dtype in (torch.float64, torch.float32):
        eng = _engine(d, W, b, dtype, "auto")
        eng.eval_numpy(d["Z"], d["X0"], want=("f", "grad", "g", "jac_dense"))
        k_dense = eng.last_row_kernel
        res = eng.eval_numpy(d["Z"], d["X0"], want=("f", "grad", "g", "jac_sparse"))
        # wherever the dense contract is one launch (fixed-shape or cooperative kernel), the sparse one is too; shapes only
        # the wave-per-tile kernel serves (fp64 3 x 128: the slices do not fit registers) keep the row + assembly launches
        expect = {"rows_coopfx_kernel": "rows_coopfx_kernel+sparse", "rows_coop_kernel+dense": "rows_coop_kernel+sparse"}
        if k_dense in expect:       # (fp32 with n % 4 != 0: no one-launch DENSE rows on the compiled shape, the band values still are)
            assert eng.last_row_kernel in (expect[k_dense], "rows_coopfx_kernel+sparse"), (name, dtype)
        if name in ("c2_discret", "c5_box", "c1_discret", "c3_rk4") and not (name == "c3_rk4" and dtype == torch.float64):
            assert k_dense in expect, (name, dtype, k_dense)
        rows, cols = eng.jac_structure()
        # every output subset of the same launch, bit for bit: values alone, with the compact tiles, with the dense matrix
        # (the dense request takes the assembly path: the band values are then gathered from the tiles)
        alone = eng.eval_numpy(d["Z"], d["X0"], want=("jac_sparse",))
        assert np.array_equal(alone["jac_sparse"], res["jac_sparse"])
        wt = eng.eval_numpy(d["Z"], d["X0"], want=("g", "jac_sparse", "jac_tiles"))
        assert np.array_equal(wt["jac_sparse"], res["jac_sparse"]) and np.array_equal(wt["g"], res["g"])
        full = eng.eval_numpy(d["Z"], d["X0"], want=ALL)
        assert np.array_equal(res["jac_sparse"], full["jac_sparse"]) and np.array_equal(res["f"], full["f"])
        assert np.array_equal(wt["jac_tiles"], full["jac_tiles"])
        assert np.array_equal(res["jac_sparse"], full["jac_dense"][:, rows, cols])
        if dtype == torch.float64:
            np.testing.assert_allclose(res["f"], d["f"], **F64)
            np.testing.assert_allclose(res["grad"], d["grad"], **F64)
            np.testing.assert_allclose(res["g"], d["g"], **F64)
            np.testing.assert_allclose(res["jac_sparse"], d["jac"][:, rows, cols], **F64)
        else:
            _f32_close(res["jac_sparse"], d["jac"][:, rows, cols], f"{name}/jac_sparse")
            _f32_close(res["f"], d["f"], f"{name}/f")


@pytest.mark.parametrize("kernel", ["valu", "mfma", "mfma_tile"])
def test_nan_inputs_stay_visible(kernel):
    """A NaN in the iterate (diverged solver) must come out as NaN in exactly the rows that read it, like the
    reference's NumPy / libm path -- not be clamped away by the fast tanh."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 6, 3
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=1)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=2)
    Zh[1, H * nx + 2] = np.nan                      # control u_2 of problem 1
    for integ, kind in (("unity", orc.UNITY), ("discret", orc.DISCRET)):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, device="cuda:0", max_batch=B, kernel=kernel)
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
        prob = orc.Problem(net, H, nx, nu, kind)
        with np.errstate(invalid="ignore"):
            gref = np.stack([prob.constraints(Zh[i], X0h[i]) for i in range(B)])
        assert np.array_equal(np.isnan(res["g"]), np.isnan(gref))
        assert np.isnan(res["g"][1, 2 * nx:3 * nx]).all() and np.isnan(res["jac_tiles"][1, 2]).all()
        ok = ~np.isnan(gref)
        np.testing.assert_allclose(res["g"][ok], gref[ok], **F64)


@pytest.mark.parametrize("kind,DT", [(orc.DISCRET, 1.0), (orc.RK4, 0.2)])
def test_linear_model_single_dense_layer(kind, DT):
    """A network with no hidden layer (one Dense-linear): the generic kernel is the only path; values, Jacobian and
    (zero-curvature) Hessian against the oracle."""
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd._lib import NempcError
    nx, nu, H, B = 3, 2, 5, 4
    net = orc.MLP.random(nx + nu, [], nx, seed=9)
    prob = orc.Problem(net, H, nx, nu, kind, DT)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=3)
    integ = {orc.DISCRET: "discret", orc.RK4: "rk4"}[kind]
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, device="cuda:0", max_batch=B)
    assert eng.kernel_variant == "valu"
    with pytest.raises(NempcError):
        CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, device="cuda:0", kernel="mfma")
    res = eng.eval_numpy(Zh, X0h)
    f, grad, g, jac = prob.eval_batch(Zh, X0h)
    np.testing.assert_allclose(res["g"], g, **F64)
    np.testing.assert_allclose(res["jac_dense"], jac, **F64)
    lam = np.random.default_rng(1).normal(size=(B, prob.m))
    hd = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.ones(B)),
                  want=("hdense",))["hdense"].cpu().numpy()
    for i in range(B):
        np.testing.assert_allclose(hd[i], prob.lagrangian_hessian(Zh[i], X0h[i], lam[i], 1.0), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("hidden", [[8] * 7, [200], [17, 33, 5]])
def test_shapes_outside_the_matrix_core_kernels(hidden):
    """Deepest allowed stack (8 dense layers), a width beyond 128 and ragged widths: generic kernel (a ragged 3-hidden
    stack pads to the matrix-core shape), against the oracle."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 2, 4, 5
    net = orc.MLP.random(nx + nu, hidden, nx, seed=4)
    prob = orc.Problem(net, H, nx, nu, orc.RK4, 0.1)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=6)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, device="cuda:0", max_batch=B)
    # (auto: what the register-resident matrix-core kernels do not take runs on the layer-at-a-time GEMM pipeline)
    assert eng.kernel_variant == ("mfma" if hidden == [17, 33, 5] else "layered")
    res = eng.eval_numpy(Zh, X0h)
    assert eng.last_row_kernel == ("layered_gemm_kernel" if eng.kernel_variant == "layered" else eng.last_row_kernel)
    f, grad, g, jac = prob.eval_batch(Zh, X0h)
    np.testing.assert_allclose(res["g"], g, **F64)
    np.testing.assert_allclose(res["jac_dense"], jac, rtol=1e-11, atol=1e-12)
    with pytest.raises(ValueError, match="at most"):
        deep = orc.MLP.random(nx + nu, [4] * 8, nx, seed=1)
        CallbackEngine(deep.W, deep.b, H, nx, nu, device="cuda:0")


def test_two_handles_on_two_streams():
    """One handle per stream (nempc.h: a handle is not re-entrant): two engines with different networks evaluated
    concurrently on their own HIP streams give the same numbers as when run alone."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 20, 200
    nets = [orc.MLP.random(nx + nu, [64, 64], nx, seed=s) for s in (1, 2)]
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=4)
    engs = [CallbackEngine(n.W, n.b, H, nx, nu, device="cuda:0", max_batch=B) for n in nets]
    Z, X0 = engs[0].to_device(Zh), engs[0].to_device(X0h)
    alone = [{k: v.clone() for k, v in e.eval(Z, X0).items()} for e in engs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in engs]
    outs = [None, None]
    for rep in range(20):
        for i, (e, st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                outs[i] = e.eval(Z, X0)
    torch.cuda.synchronize()
    for i in range(2):
        for k in alone[i]:
            assert torch.equal(outs[i][k], alone[i][k]), (i, k)
    assert not torch.equal(outs[0]["g"], outs[1]["g"])


@pytest.mark.parametrize("name,kernel", [(n, k) for n in ["c2_discret", "c5_box", "tvp_p_discret"] +
                                         [c for c in ACT_UNIFORM_NAMES if c.endswith("_c2")]
                                         for k in ("mfma", "mfma_tile", "valu")] + [("act_mixed_box", "valu")])
def test_golden_hessian_fp32(name, kernel):
    """fp32 Lagrangian Hessian (cooperative, wave-per-tile and generic kernels) within 1e-4 of the fp64 golden."""
    d, W, b = load_case(name)
    eng = _engine(d, W, b, torch.float32, kernel)
    out = eng.hess(eng.to_device(d["Z"]), eng.to_device(d["X0"]), eng.to_device(d["lam"]), eng.to_device(d["sigma"]),
                   want=("hdense",))
    hd = out["hdense"].cpu().numpy().astype(np.float64)
    _f32_close(hd, d["hdense"], f"{name}/{kernel}/hdense")
    assert np.array_equal(hd, np.transpose(hd, (0, 2, 1)))


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
def test_terminal_weight(kernel):
    """Terminal cost: the last step's state weight QT in f, grad f, the Lagrangian Hessian and the batched solver."""
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 9, 6
    net = orc.MLP.random(nx + nu, [32, 32], nx, seed=2)
    net.W[-1] *= 0.2
    net.b[-1] *= 0.2
    rng = np.random.default_rng(5)
    Q, R, QT = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx)), 0.2 * np.eye(nu), 5.0 * np.eye(nx) + rng.normal(size=(nx, nx))
    QT = QT @ QT.T / 4.0
    xref = rng.normal(size=(H, nx)) * 0.2
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=Q, R=R, xref=xref, QT=QT)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=3)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B, kernel=kernel)
    eng.set_objective(Q=Q, R=R, xref=xref, QT=QT)
    res = eng.eval_numpy(Zh, X0h, want=("f", "grad"))
    np.testing.assert_allclose(res["f"], [prob.objective(z) for z in Zh], **F64)
    np.testing.assert_allclose(res["grad"], [prob.gradient(z) for z in Zh], **F64)
    lam = rng.normal(size=(B, prob.m))
    hd = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.full(B, 0.7)),
                  want=("hdense",))["hdense"].cpu().numpy()
    for i in range(B):
        np.testing.assert_allclose(hd[i], prob.lagrangian_hessian(Zh[i], X0h[i], lam[i], 0.7), rtol=1e-11, atol=1e-12)
    # solver: KKT point of the oracle's problem with the terminal weight
    Z, st, _ = eng.solve(eng.to_device(X0h), max_iter=80)
    Z = Z.cpu().numpy()
    assert int((st == 0).sum().item()) >= B - 1
    for i in np.nonzero(st.cpu().numpy() == 0)[0][:3]:
        assert np.abs(prob.constraints(Z[i], X0h[i])).max() < 1e-7
        J, gr = prob.jacobian(Z[i], X0h[i]), prob.gradient(Z[i])
        mult = np.linalg.lstsq(J.T, -gr, rcond=None)[0]
        assert np.abs(gr + J.T @ mult).max() < 1e-5 * max(1.0, np.abs(gr).max())
    # back to the plain family
    eng.set_objective(Q=Q, R=R, xref=xref)
    plain = orc.Problem(net, H, nx, nu, orc.DISCRET, Q=Q, R=R, xref=xref)
    np.testing.assert_allclose(eng.eval_numpy(Zh, X0h, want=("f",))["f"], [plain.objective(z) for z in Zh], **F64)


def test_eval_pipeline_keeps_independent_batches_in_flight():
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd.parallel import EvalPipeline
    nx, nu, H, B = 2, 1, 20, 64
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=0)
    make = lambda: CallbackEngine(net.W, net.b, H, nx, nu, device="cuda:0", max_batch=B)
    ref = make()
    pipe = EvalPipeline(make, depth=2)
    batches = [tuple(ref.to_device(a) for a in orc.synthetic_inputs(B, H, nx, nu, seed=10 + k)) for k in range(6)]
    expect = [{k: v.clone() for k, v in ref.eval(Z, X0).items()} for Z, X0 in batches]
    tickets = []
    for k, (Z, X0) in enumerate(batches):
        if k >= 2:                                   # consume the slot's previous result before reusing it
            out = pipe.wait(tickets[k - 2])
            for name in out:
                assert torch.equal(out[name], expect[k - 2][name])
        tickets.append(pipe.submit(Z, X0))
    for k in (4, 5):
        out = pipe.wait(tickets[k])
        torch.cuda.current_stream().synchronize()
        for name in out:
            assert torch.equal(out[name], expect[k][name])
    pipe.synchronize()
    with pytest.raises(ValueError):
        EvalPipeline(make, depth=0)



@pytest.mark.parametrize("act", ["relu", "sigmoid", "softplus", "elu"])
def test_activation_family_seeded_against_oracle(act):
    """One activation on every hidden layer: all three kernel families against the oracle at shapes the goldens do not
    hold -- the fixed-shape one-launch path (2/1, 2x64, B*H not a multiple of 16, with box rows), RK4 at 6/3 3x128, the
    RK4 Lagrangian Hessian pipeline, the Gauss-Newton callback -- and against each other."""
    from pyneuralempc_amd import CallbackEngine
    rng = np.random.default_rng(12)
    for nx, nu, hidden, H, kind, DT, box, B in ((2, 1, [64, 64], 20, orc.DISCRET, 1.0, (-2.0, 2.0), 37),
                                                (6, 3, [128, 128, 128], 5, orc.RK4, 0.1, None, 7),
                                                (3, 2, [48, 32], 4, orc.RK4, 0.2, None, 5)):
        net = orc.MLP.random(nx + nu, hidden, nx, seed=3, activations=act)
        prob = orc.Problem(net, H, nx, nu, kind, DT, box=box)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=6)
        lamh, sigh = rng.normal(size=(B, prob.m)), rng.uniform(0.5, 1.5, size=B)
        wh = rng.uniform(0.2, 1.5, size=(B, H * nx))
        f, grad, g, jac = prob.eval_batch(Zh, X0h)
        hv = np.stack([prob.hessian_values(Zh[i], X0h[i], lamh[i], sigh[i]) for i in range(B)])
        gn = np.stack([prob.gauss_newton_values(Zh[i], X0h[i], wh[i], sigh[i]) for i in range(B)])
        outs = {}
        for kernel in ("valu", "mfma", "mfma_tile"):
            eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=KIND_NAME[kind], DT=DT, dtype=torch.float64,
                                 device="cuda:0", max_batch=B, kernel=kernel, activations=act)
            assert eng.kernel_variant == kernel
            if box is not None:
                eng.set_box_rows(*box)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            res = {k: v.cpu().numpy() for k, v in eng.eval(Z, X0, ("f", "grad", "g", "jac_dense")).items()}
            if kernel == "mfma" and nx == 2:
                assert eng.last_row_kernel == "rows_coopfx_kernel"
            np.testing.assert_allclose(res["f"], f, **F64)
            np.testing.assert_allclose(res["grad"], grad, **F64)
            np.testing.assert_allclose(res["g"], g, **F64)
            np.testing.assert_allclose(res["jac_dense"], jac, **F64)
            assert np.array_equal(res["jac_dense"] != 0, jac != 0) or act == "relu"    # (a dead relu row is an exact 0)
            h = eng.hess(Z, X0, eng.to_device(lamh), eng.to_device(sigh))["hvals"].cpu().numpy()
            np.testing.assert_allclose(h, hv, rtol=1e-10, atol=1e-11)
            hg = eng.hess_gn(Z, X0, eng.to_device(wh), eng.to_device(sigh))["hvals"].cpu().numpy()
            np.testing.assert_allclose(hg, gn, rtol=1e-11, atol=1e-12)
            outs[kernel] = res
        np.testing.assert_allclose(outs["mfma"]["jac_dense"], outs["valu"]["jac_dense"], **F64)


def test_mixed_activations_run_on_the_layered_path_and_are_refused_by_the_register_resident_kernels():
    from pyneuralempc_amd import CallbackEngine, _lib
    net = orc.MLP.random(3, [32, 32], 2, seed=1, activations=["relu", "tanh", "linear"])
    eng = CallbackEngine(net.W, net.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=net.act)
    # (round 5) a per-layer mix of output-based activations under a linear output layer: the register-resident kernels with
    # run-time activation codes
    assert eng.kernel_variant == "mfma"
    Zh, X0h = orc.synthetic_inputs(4, 6, 2, 1, seed=2)
    res = eng.eval_numpy(Zh, X0h)
    f, grad, g, jac = orc.Problem(net, 6, 2, 1).eval_batch(Zh, X0h)
    np.testing.assert_allclose(res["jac_dense"], jac, **F64)
    np.testing.assert_allclose(res["g"], g, **F64)
    # fp64 mixes whose slices do not fit the cooperative kernel's registers: AUTO takes the layered path (measured faster),
    # the kernels remain available by name; the same shape in fp32 stays register-resident
    big = orc.MLP.random(3, [128, 128, 128], 2, seed=1, activations=["relu", "tanh", "sigmoid", "linear"])
    assert CallbackEngine(big.W, big.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=big.act).kernel_variant == "layered"
    assert CallbackEngine(big.W, big.b, 6, 2, 1, dtype=torch.float32, device="cuda:0", max_batch=4, activations=big.act).kernel_variant == "mfma"
    assert CallbackEngine(big.W, big.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=big.act,
                          kernel="mfma").kernel_variant == "mfma"
    # a non-linear OUTPUT layer is outside them (the layered path takes it), and asking for them by name says so
    nso = orc.MLP.random(3, [32, 32], 2, seed=1, activations=["relu", "tanh", "sigmoid"])
    assert CallbackEngine(nso.W, nso.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=nso.act).kernel_variant == "layered"
    with pytest.raises(_lib.NempcError, match="activations"):
        CallbackEngine(nso.W, nso.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", kernel="mfma", activations=nso.act)
    with pytest.raises(NotImplementedError, match="hard_sigmoid"):
        CallbackEngine(net.W, net.b, 6, 2, 1, device="cuda:0", activations=["hard_sigmoid", "tanh", "linear"])
    # swish / gelu ... (derivatives from the pre-activation): the layered path, the generic kernel by name, any layer -- the
    # output layer included (round 5); never the register-resident kernels
    assert CallbackEngine(net.W, net.b, 6, 2, 1, device="cuda:0", activations=["swish", "gelu", "linear"]).kernel_variant == "layered"
    assert CallbackEngine(net.W, net.b, 6, 2, 1, device="cuda:0", kernel="valu", activations=["swish", "gelu", "linear"]).kernel_variant == "valu"
    zo = CallbackEngine(net.W, net.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=["tanh", "tanh", "gelu"])
    assert zo.kernel_variant == "layered"
    nzo = orc.MLP(net.W, net.b, ["tanh", "tanh", "gelu"])
    np.testing.assert_allclose(zo.eval_numpy(Zh, X0h)["jac_dense"], orc.Problem(nzo, 6, 2, 1).eval_batch(Zh, X0h)[3], **F64)
    with pytest.raises(_lib.NempcError, match="activations"):
        CallbackEngine(net.W, net.b, 6, 2, 1, device="cuda:0", kernel="mfma", activations=["swish", "gelu", "linear"])
    with pytest.raises(ValueError, match="one name per dense layer"):
        CallbackEngine(net.W, net.b, 6, 2, 1, device="cuda:0", activations=["tanh", "linear"])
    # a non-linear OUTPUT layer with one hidden activation is a mix too
    n2 = orc.MLP.random(3, [32], 2, seed=1, activations=["tanh", "sigmoid"])
    e2 = CallbackEngine(n2.W, n2.b, 6, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, activations=n2.act)
    assert e2.kernel_variant == "layered"
    np.testing.assert_allclose(e2.eval_numpy(Zh, X0h)["jac_dense"], orc.Problem(n2, 6, 2, 1).eval_batch(Zh, X0h)[3], **F64)


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("shape", ["3x128_tanh", "2x128_mix"])
def test_fp64_three_by_128_rows_stay_register_resident_and_the_lagrangian_blocks_take_the_layered_sweeps(kind, shape):
    """AUTO, fp64, width 128 (kernels_mfma.hip: mfma_hess_on_layered) -- three hidden layers with one activation, two with a
    per-layer mix: the rows launch is the register-resident kernel, the exact Hessian the layer-at-a-time sweeps (measured
    faster there); both against the oracle and against the register-resident Hessian kernel asked for by name; hvals / dense /
    blocks outputs."""
    from pyneuralempc_amd import CallbackEngine
    B, H, nx, nu = 37, 9, 3, 2
    hidden, acts, rowk, hessk = (([128, 128, 128], None, "rows_mfma_kernel", "rowhess_mfma_kernel") if shape == "3x128_tanh" else
                                 ([128, 128], ["tanh", "softplus", "linear"], "rows_coop_kernel", "rowhess_coop_kernel"))
    net = orc.MLP.random(nx + nu, hidden, nx, seed=4, activations=acts)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    rng = np.random.default_rng(6)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=KIND_NAME[kind], DT=0.1, dtype=torch.float64, device="cuda:0", max_batch=B,
                         activations=net.act)
    byname = CallbackEngine(net.W, net.b, H, nx, nu, integrator=KIND_NAME[kind], DT=0.1, dtype=torch.float64, device="cuda:0",
                            max_batch=B, kernel="mfma", activations=net.act)
    assert eng.kernel_variant == "mfma" and byname.kernel_variant == "mfma"
    lamh, sigh = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 2.0, size=B)
    Z, X0, lam, sig = (eng.to_device(a) for a in (Zh, X0h, lamh, sigh))
    res = eng.eval(Z, X0, ("g", "jac_dense"))
    assert eng.last_row_kernel == rowk      # (fp64 slices of 3 x 128 do not fit the cooperative kernel)
    prob = orc.Problem(net, H, nx, nu, kind, 0.1)
    _, _, g, jac = prob.eval_batch(Zh, X0h)
    np.testing.assert_allclose(res["jac_dense"].cpu().numpy(), jac, **F64)
    for want in (("hvals",), ("hvals", "hdense", "hblocks")):
        out = eng.hess(Z, X0, lam, sig, want=want)
        assert eng.last_hess_kernel == "layered_gemm_kernel"
        ref = byname.hess(Z, X0, lam, sig, want=want)
        assert byname.last_hess_kernel == hessk
        for k in want:
            np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-10, atol=1e-11)
    hd = out["hdense"].cpu().numpy()
    for i in (0, B - 1):
        np.testing.assert_allclose(hd[i], prob.lagrangian_hessian(Zh[i], X0h[i], lamh[i], sigh[i]), rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid", "softplus"])
def test_nan_inputs_stay_visible_for_every_activation(act):
    from pyneuralempc_amd import CallbackEngine
    net = orc.MLP.random(3, [64, 64], 2, seed=0, activations=act)
    for kernel in ("valu", "mfma", "mfma_tile"):
        eng = CallbackEngine(net.W, net.b, 20, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=4, kernel=kernel,
                             activations=act)
        Zh, X0h = orc.synthetic_inputs(4, 20, 2, 1, seed=1)
        Zh[1, 5] = np.nan
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
        assert np.isnan(res["g"][1]).any() and not np.isnan(res["g"][[0, 2, 3]]).any(), (act, kernel)


def test_every_matrix_core_row_instantiation_against_the_oracle():
    """Every (dtype, padded width 32 / 64 / 128, hidden layers 1..3) instantiation of the wave-per-tile row kernel x activation
    x transcription once, defects and dense Jacobian against the oracle.  Round 3 found three streamed instantiations at the
    register cap returning wrong rows for some of these combinations and kept them away from the launcher; round 4 found the
    cause (a spill store in front of an exec restore, pyneuralempc_amd/_isa.py), the build repairs it, and every
    instantiation is launched again: nothing may fall back to the generic kernel here."""
    import itertools
    from pyneuralempc_amd import CallbackEngine
    nx, nu, H, B = 2, 1, 7, 37
    for dt, width, depth, integ, act in itertools.product((torch.float64, torch.float32), (24, 48, 96), (1, 2, 3),
                                                         ("discret", "rk4"), ("tanh", "relu", "sigmoid", "softplus", "elu")):
        DT = 0.1 if integ == "rk4" else 1.0
        net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=5, activations=act)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
        prob = orc.Problem(net, H, nx, nu, orc.RK4 if integ == "rk4" else orc.DISCRET, DT)
        f, grad, g, J = prob.eval_batch(Zh, X0h)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B,
                             kernel="mfma_tile", activations=act)
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        tol = 2e-4 if dt == torch.float32 else 1e-10
        tag = (str(dt), width, depth, integ, act)
        assert eng.last_row_kernel == "rows_mfma_kernel", tag
        assert np.abs(res["g"] - g).max() / max(1.0, np.abs(g).max()) < tol, tag
        assert np.abs(res["jac_dense"] - J).max() / max(1.0, np.abs(J).max()) < tol, tag
        # defect-only launches (line-search trials) take the wave-per-tile kernel on every variant
        g_only = eng.eval_numpy(Zh, X0h, want=("g",))["g"]
        assert np.abs(g_only - g).max() / max(1.0, np.abs(g).max()) < tol, tag
        del eng


def test_every_matrix_core_instantiation_with_its_hessian_against_the_oracle():
    """The same sweep over the kernels the DEFAULT dispatch launches (`kernel="mfma"`: cooperative rows, cooperative /
    fixed-shape / wave-per-tile Hessian kernels, the RK4 Hessian pipeline) and the wave-per-tile family, rows AND Lagrangian
    Hessian values, 2/1 and 6/3 dims: every launched instantiation that sits at the 256-register cap or has scratch
    (`rows_coop_kernel<.., 64, 3, ..>`, `<.., 128, 2, ..>`, `rowhess_coop_kernel<.., 64, 3, 3>`, `rowhess_mfma_kernel<.., 128,
    {2, 3}, false>`) is in it, each activation."""
    import itertools
    from pyneuralempc_amd import CallbackEngine
    H, B = 7, 11
    for dt, width, depth, integ, act, kern, (nx, nu) in itertools.product(
            (torch.float64, torch.float32), (48, 96), (2, 3), ("discret", "rk4"), ("tanh", "relu", "sigmoid", "softplus", "elu"),
            ("mfma", "mfma_tile"), ((2, 1), (6, 3))):
        DT = 0.1 if integ == "rk4" else 1.0
        net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=5, activations=act)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
        prob = orc.Problem(net, H, nx, nu, orc.RK4 if integ == "rk4" else orc.DISCRET, DT)
        f, grad, g, J = prob.eval_batch(Zh, X0h)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B,
                             kernel=kern, activations=act)
        tag = (str(dt), width, depth, integ, act, kern, nx)
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        assert eng.last_row_kernel != "rows_valu_kernel", tag
        tol = 2e-4 if dt == torch.float32 else 1e-10
        assert np.abs(res["g"] - g).max() / max(1.0, np.abs(g).max()) < tol, tag
        assert np.abs(res["jac_dense"] - J).max() / max(1.0, np.abs(J).max()) < tol, tag
        rng = np.random.default_rng(1)
        lam, sig = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
        hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(sig))["hvals"].cpu().double().numpy()
        ref = np.stack([prob.hessian_values(Zh[i], X0h[i], lam[i], sig[i]) for i in range(B)])
        assert np.abs(hv - ref).max() / max(1.0, np.abs(ref).max()) < (2e-3 if dt == torch.float32 else 1e-9), tag
        del eng



@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("hidden,acts,nx,nu,integ", [
    ([256, 256], "tanh", 2, 1, "discret"),                                   # wide256_c2
    ([96, 96, 96, 96], "tanh", 2, 1, "discret"),                            # four hidden layers beyond width 64
    ([200, 130, 70], ["relu", "sigmoid", "elu", "softplus"], 3, 2, "unity"),  # ragged widths, a mix, non-linear output
    ([144, 96, 96, 40, 24], ["tanh", "relu", "tanh", "softplus", "elu", "linear"], 6, 3, "rk4"),
    ([512], "sigmoid", 1, 1, "rk4"),
    ([300], "tanh", 2, 1, "discret"),                                        # one hidden layer: no tangent products at all
    ([160, 160], ["selu", "leaky_relu:0.1", "linear"], 4, 2, "discret"),
    ([192, 130], ["swish", "gelu", "linear"], 2, 1, "discret"),             # derivatives from the pre-activation
    ([96, 96, 96], ["gelu", "tanh", "swish", "softplus"], 3, 2, "rk4"),
    ([150], ["swish", "linear"], 2, 2, "unity"),
    ([140, 90], ["mish", "softsign", "linear"], 2, 1, "discret"),
    ([100, 100, 60], ["exponential", "relu6", "mish", "tanh"], 3, 1, "rk4"),
])
def test_layered_matrix_core_path_against_the_oracle(dtype, hidden, acts, nx, nu, integ):
    """Networks outside the register-resident kernels (width > 128, more than three hidden layers, per-layer activation
    mixes): one GEMM launch per layer on the matrix cores (csrc/kernels_layered.hip) -- every contract (g, dense, sparse,
    tiles, objective), ragged batches incl. a chunk boundary inside a GEMM block, box rows -- against the oracle, and bit
    for bit independent of the batch a row sits in."""
    from pyneuralempc_amd import CallbackEngine
    H = 7
    DT = 0.1 if integ == "rk4" else 1.0
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[integ]
    net = orc.MLP.random(nx + nu, hidden, nx, seed=8, activations=acts)
    box = (-1.5, 1.5) if integ == "discret" else None
    prob = orc.Problem(net, H, nx, nu, kind, DT, box=box)
    f64 = dtype == torch.float64
    tol = dict(rtol=1e-11, atol=1e-11) if f64 else dict(rtol=3e-4, atol=3e-4)
    ref_rows = ref_hess = None
    for B in (1, 19, 150):
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dtype, device="cuda:0", max_batch=B,
                             activations=net.act)
        assert eng.kernel_variant == "layered"
        if box:
            eng.set_box_rows(*box)
        Zh, X0h = orc.synthetic_inputs(150, H, nx, nu, seed=3)
        Zh, X0h = Zh[:B], X0h[:B]
        res = eng.eval_numpy(Zh, X0h, want=ALL)
        assert eng.last_row_kernel == "layered_gemm_kernel"
        k = min(B, 12)
        f, grad, g, J = prob.eval_batch(Zh[:k], X0h[:k])
        np.testing.assert_allclose(res["f"][:k], f, **tol)
        np.testing.assert_allclose(res["grad"][:k], grad, **tol)
        np.testing.assert_allclose(res["g"][:k], g, **tol)
        np.testing.assert_allclose(res["jac_dense"][:k], J, **tol)
        rows, cols = eng.jac_structure()
        assert np.array_equal(res["jac_sparse"], res["jac_dense"][:, rows, cols])
        assert np.array_equal(res["jac_dense"] != 0, (np.abs(res["jac_dense"]) > 0)) and np.isfinite(res["jac_dense"]).all()
        only_g = eng.eval_numpy(Zh, X0h, want=("g",))["g"]
        assert np.array_equal(only_g, res["g"])
        if B == 150:
            # the Lagrangian blocks too: bit for bit what the same problems gave in the batch of 19, and the same again when
            # the callback is repeated (fixed summation order in every sweep)
            lam150 = np.random.default_rng(1).normal(size=(B, eng.m))
            lam150[:7] = ref_hess[1]
            args = (eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam150), eng.to_device(np.ones(B)))
            h1 = eng.hess(*args)["hvals"].to("cpu", torch.float64).numpy()
            h2 = eng.hess(*args)["hvals"].to("cpu", torch.float64).numpy()
            assert np.array_equal(h1, h2)
            assert np.array_equal(h1[:7], ref_hess[0])
        if ref_rows is None:
            ref_rows = res["jac_tiles"][0].copy()
        else:
            assert np.array_equal(res["jac_tiles"][0], ref_rows)       # problem 0 does not depend on the batch around it
        # generic kernel of the same handle shape: agreement to rounding (swish / gelu exist on the layered path only)
        zbased = any(str(a).split(":")[0] in orc.ZBASED for a in net.act)
        if B == 19:
            ev = None
            if not zbased:
                ev = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dtype, device="cuda:0", max_batch=B,
                                    activations=net.act, kernel="valu")
                if box:
                    ev.set_box_rows(*box)
                rv = ev.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
                assert ev.last_row_kernel == "rows_valu_kernel"
                np.testing.assert_allclose(res["g"], rv["g"], **tol)
                np.testing.assert_allclose(res["jac_tiles"], rv["jac_tiles"], **tol)
            # the Lagrangian Hessian of such a model: the GEMM sweeps with the layer-wise contraction (csrc/kernels_layered.hip,
            # "Contracted network Hessian"), for RK4 inside the stage pipeline of kernels_rk4hess.hip -- against the oracle
            # and against the generic kernel
            lam = np.random.default_rng(1).normal(size=(B, eng.m))
            hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.ones(B)))["hvals"]
            assert eng.last_hess_kernel == ("rk4:layered_gemm_kernel" if integ == "rk4" else "layered_gemm_kernel")
            hv = hv.to("cpu", torch.float64).numpy()
            for i in range(3):
                refh = prob.hessian_values(Zh[i], X0h[i], lam[i], 1.0)
                np.testing.assert_allclose(hv[i], refh, rtol=0, atol=(1e-9 if f64 else 5e-3) * max(1.0, np.abs(refh).max()))
            ref_hess = (hv[:7].copy(), lam[:7].copy())
            if ev is not None:
                hg = ev.hess(ev.to_device(Zh), ev.to_device(X0h), ev.to_device(lam), ev.to_device(np.ones(B)))["hvals"]
                assert ev.last_hess_kernel == "rowhess_valu_kernel"
                hg = hg.to("cpu", torch.float64).numpy()
                np.testing.assert_allclose(hv, hg, rtol=0, atol=(1e-10 if f64 else 5e-3) * max(1.0, np.abs(hg).max()))


@pytest.mark.parametrize("hidden,nx,nu,integ,H,B", [
    ([1024, 1000], 16, 16, "discret", 3, 3),         # the widest layers, 32 network inputs, 16 cotangents
    ([40] * 7, 2, 1, "rk4", 4, 5),                   # eight dense layers
    ([72, 72], 3, 1, "unity", 1, 1),                 # one row
    ([130], 5, 4, "rk4", 2, 70),                     # one hidden layer, more rows than one GEMM block
])
def test_layered_path_at_the_edges_of_its_shape_range(hidden, nx, nu, integ, H, B):
    """The layered path's limits (widths <= 1024, <= 8 layers, <= 32 network inputs, nx <= 16) and its smallest launches:
    rows and Lagrangian blocks against the oracle."""
    from pyneuralempc_amd import CallbackEngine
    DT = 0.1 if integ == "rk4" else 1.0
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[integ]
    net = orc.MLP.random(nx + nu, hidden, nx, seed=21)
    prob = orc.Problem(net, H, nx, nu, kind, DT)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B,
                         kernel="layered")
    assert eng.kernel_variant == "layered"
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=4)
    res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
    k = min(B, 3)
    f, grad, g, J = prob.eval_batch(Zh[:k], X0h[:k])
    np.testing.assert_allclose(res["g"][:k], g, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(res["jac_dense"][:k], J, rtol=1e-10, atol=1e-10)
    lam = np.random.default_rng(2).normal(size=(B, eng.m))
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.ones(B)))["hvals"].cpu().numpy()
    assert eng.last_hess_kernel.endswith("layered_gemm_kernel")
    for i in range(min(B, 2)):
        ref = prob.hessian_values(Zh[i], X0h[i], lam[i], 1.0)
        np.testing.assert_allclose(hv[i], ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("hidden,acts,nx,nu,integ", [
    ([48], "tanh", 2, 1, "discret"),                    # K = nin <= 16: ONE chunk in every product
    ([80, 48], "tanh", 2, 1, "discret"),                # 5 and 3 chunks
    ([48, 80, 48], ["tanh", "sigmoid", "tanh", "linear"], 3, 2, "rk4"),
    ([144, 176], ["swish", "gelu", "linear"], 2, 1, "unity"),   # 9 and 11 chunks; pre-activation derivatives
])
def test_layered_products_with_an_odd_number_of_chunks_on_a_full_chip(hidden, acts, nx, nu, integ):
    """A product whose K has an odd number of 16-deep chunks ends on `mma_chunk(0)` with no barrier behind it; the fused
    contractions then write their per-wave partial tiles into the operand buffers (round-4 advisor: a wave that finishes
    early overwrote operands slower waves were still reading).  Enough rows that every SIMD holds several waves, every
    row of the batch compared -- forward contraction, reverse contraction (J) and the Hessian sweeps -- with the generic
    kernel where it exists and with the oracle on a slice, and twice bit for bit."""
    from pyneuralempc_amd import CallbackEngine
    H, B = 20, 1536
    DT = 0.1 if integ == "rk4" else 1.0
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[integ]
    net = orc.MLP.random(nx + nu, hidden, nx, seed=13, activations=acts)
    prob = orc.Problem(net, H, nx, nu, kind, DT)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B,
                         activations=net.act, kernel="layered")
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
    lam = np.random.default_rng(3).normal(size=(B, eng.m))
    dev = [eng.to_device(a) for a in (Zh, X0h, lam, np.ones(B))]
    r1 = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
    h1 = eng.hess(*dev)["hvals"].cpu().numpy()
    for _ in range(3):
        r2 = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
        assert np.array_equal(r1["g"], r2["g"]) and np.array_equal(r1["jac_tiles"], r2["jac_tiles"])
        assert np.array_equal(h1, eng.hess(*dev)["hvals"].cpu().numpy())
    if not any(str(a).split(":")[0] in orc.ZBASED for a in net.act):
        ev = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B,
                            activations=net.act, kernel="valu")
        rv = ev.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
        np.testing.assert_allclose(r1["g"], rv["g"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(r1["jac_tiles"], rv["jac_tiles"], rtol=1e-11, atol=1e-11)
        hg = ev.hess(*[ev.to_device(a) for a in (Zh, X0h, lam, np.ones(B))])["hvals"].cpu().numpy()
        np.testing.assert_allclose(h1, hg, rtol=0, atol=1e-10 * max(1.0, np.abs(hg).max()))
    for s0 in (0, B // 2, B - 2):
        f, grad, g, J = prob.eval_batch(Zh[s0:s0 + 2], X0h[s0:s0 + 2])
        np.testing.assert_allclose(r1["g"][s0:s0 + 2], g, rtol=1e-11, atol=1e-11)
        ref = prob.hessian_values(Zh[s0], X0h[s0], lam[s0], 1.0)
        np.testing.assert_allclose(h1[s0], ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


_AB_WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
out = {}
for tag, hidden, acts, nx, nu, integ in (("a", [200, 136], ["tanh", "sigmoid", "linear"], 2, 1, "discret"), ("b", [144], "tanh", 3, 2, "rk4"),
                                         ("c", [136, 150, 72], ["tanh", "softplus", "relu", "linear"], 2, 2, "rk4"),
                                         ("d", [160, 130], "tanh", 1, 1, "unity")):
    H, B, DT = 6, 90, (0.1 if integ == "rk4" else 1.0)
    net = orc.MLP.random(nx + nu, hidden, nx, seed=5, activations=acts)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=torch.float64, device="cuda:0", max_batch=B,
                         activations=net.act, kernel="layered")
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=6)
    r = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))
    lam = np.random.default_rng(2).normal(size=(B, eng.m))
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), eng.to_device(lam), eng.to_device(np.ones(B)))["hvals"].cpu().numpy()
    out[tag + "_g"], out[tag + "_J"], out[tag + "_H"] = r["g"], r["jac_tiles"], hv
    out[tag + "_hk"] = np.array([eng.last_hess_kernel])
np.savez(sys.argv[2], **out)
"""


def test_layered_path_run_time_switches_agree_with_the_default(tmp_path):
    """The layered path's run-time switches -- NEMPC_LAYERED_FUSE=0 (seed kernel, plain products, skinny last steps: also the
    reverse sweep every single-hidden-layer network takes), NEMPC_LG_RM=2 / 4 and NEMPC_LG_RM_REV=2 (32- / 64-row blocks),
    NEMPC_LAYERED_HESS=0 (Lagrangian blocks from the generic kernel), NEMPC_LAYERED_DFA=0 / 2 (every layer stores s' / s'' next to
    its activation; or none that can form them from it does, in the rows path too), NEMPC_LAYERED_HFOLD=0 (the Hessian's
    tangents through memory and layered_hcontract_kernel instead of the tangent product's epilogue), NEMPC_LG_COT_ORDER=0 (the
    reverse products' workgroups cotangent-major instead of row-block-major), NEMPC_LAYERED_FIRST=0 (gather launch + a one-chunk
    product for layer 0 instead of layered_first_kernel), NEMPC_LAYERED_OUTSKIP=0 (a linear output layer's partial sums through
    layered_outfinish_kernel instead of the finish kernel) -- are read once per process: each runs in a process of
    its own and has to reproduce the default's rows and Lagrangian blocks to rounding (round-4 review: switches nobody tests
    are build variants nobody knows)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_AB_WORKER)
    res = {}
    for name, env in (("default", {}), ("nofuse", {"NEMPC_LAYERED_FUSE": "0"}), ("rm2", {"NEMPC_LG_RM": "2"}), ("rm4", {"NEMPC_LG_RM": "4"}),
                      ("rmrev2", {"NEMPC_LG_RM_REV": "2"}), ("nohess", {"NEMPC_LAYERED_HESS": "0"}),
                      ("nodfa", {"NEMPC_LAYERED_DFA": "0"}), ("dfa_all", {"NEMPC_LAYERED_DFA": "2"}), ("nohfold", {"NEMPC_LAYERED_HFOLD": "0"}),
                      ("dfa_all_nofuse", {"NEMPC_LAYERED_DFA": "2", "NEMPC_LAYERED_FUSE": "0"}),
                      ("cot_major", {"NEMPC_LG_COT_ORDER": "0"}), ("nofirst", {"NEMPC_LAYERED_FIRST": "0"}),
                      ("nofirst_nodfa", {"NEMPC_LAYERED_FIRST": "0", "NEMPC_LAYERED_DFA": "0"}), ("outstep", {"NEMPC_LAYERED_OUTSKIP": "0"})):
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, str(script), repo, str(out)], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, (name, r.stderr[-1500:])
        res[name] = dict(np.load(out))
    ref = res["default"]
    for name, got in res.items():
        for k in [t + sfx for t in "abcd" for sfx in ("_g", "_J", "_H")]:
            np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=1e-11 * max(1.0, np.abs(ref[k]).max()), err_msg=f"{name}/{k}")
    assert str(ref["a_hk"][0]) == "layered_gemm_kernel" and str(res["nohess"]["a_hk"][0]) == "rowhess_valu_kernel"    # (the switch did switch)


def test_layered_path_chunks_large_batches():
    """More rows than one workspace chunk holds (NEMPC_LAYERED_CHUNK_ROWS shrinks the chunk for the test): the chunk loop,
    with a last chunk that is not a whole GEMM block, gives the rows of the one-chunk evaluation bit for bit."""
    import os
    from pyneuralempc_amd import CallbackEngine
    H, nx, nu, B = 5, 2, 1, 77
    net = orc.MLP.random(nx + nu, [160, 160], nx, seed=2)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=3)
    outs = []
    for chunk in ("0", "128"):
        os.environ["NEMPC_LAYERED_CHUNK_ROWS"] = chunk
        try:
            eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
            outs.append(eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles")))
        finally:
            del os.environ["NEMPC_LAYERED_CHUNK_ROWS"]
    assert np.array_equal(outs[0]["g"], outs[1]["g"]) and np.array_equal(outs[0]["jac_tiles"], outs[1]["jac_tiles"])
    f, grad, g, J = orc.Problem(net, H, nx, nu).eval_batch(Zh[:8], X0h[:8])
    np.testing.assert_allclose(outs[1]["g"][:8], g, **F64)
