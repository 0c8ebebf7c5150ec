"""GPU: the Gauss-Newton Hessian callback (nempc_hess_gn; BASELINE.json north_star / configs[4]) against the oracle.
The reference's own Hessian (optimizer/ipopt.py:66-86) is the exact Lagrangian one -- tested in test_gpu_parity.py; this
is the first-order variant in the same sparsity pattern."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc
from helpers import load_case, oracle_problem

pytestmark = pytest.mark.gpu
KIND_NAME = {0: "discret", 1: "unity", 2: "rk4"}


def _engine(net, H, nx, nu, B, kind="discret", DT=1.0, kernel="auto", dtype=torch.float64, box=None):
    from pyneuralempc_amd import CallbackEngine
    eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=kind, DT=DT, dtype=dtype, device="cuda:0", max_batch=B,
                         kernel=kernel)
    if box is not None:
        eng.set_box_rows(*box)
    return eng


@pytest.mark.parametrize("kernel", ["mfma", "mfma_tile", "valu", "layered"])
@pytest.mark.parametrize("shape", [(2, 1, [64, 64], 50, "discret", 1.0, (-2.0, 2.0)),      # configs[4] dims
                                   (2, 1, [64, 64], 20, "rk4", 0.1, None),
                                   (3, 2, [24, 40], 7, "unity", 1.0, None)])
def test_gauss_newton_hessian_against_oracle(shape, kernel):
    nx, nu, hidden, H, kind, DT, box = shape
    B = 9
    net = orc.MLP.random(nx + nu, hidden, nx, seed=2)
    eng = _engine(net, H, nx, nu, B, kind, DT, kernel, box=box)
    Q = np.eye(nx) + 0.1 * np.arange(nx * nx).reshape(nx, nx)
    eng.set_objective(Q=Q, R=0.2 * np.eye(nu), QT=2.0 * np.eye(nx))
    prob = orc.Problem(net, H, nx, nu, {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[kind], DT, Q=Q,
                       R=0.2 * np.eye(nu), QT=2.0 * np.eye(nx), box=box)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=6)
    rng = np.random.default_rng(0)
    wh = rng.uniform(0.1, 2.0, size=(B, H * nx))
    sg = rng.uniform(0.5, 1.5, size=B)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    out = eng.hess_gn(Z, X0, eng.to_device(wh), eng.to_device(sg), want=("hvals", "hdense", "hblocks"))
    hv, hd = out["hvals"].cpu().numpy(), out["hdense"].cpu().numpy()
    rows, cols = eng.hess_structure()
    for b in range(B):
        ref = prob.gauss_newton_hessian(Zh[b], X0h[b], wh[b], sg[b])
        np.testing.assert_allclose(hd[b], ref, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(hv[b], ref[rows, cols], rtol=1e-11, atol=1e-11)
        assert np.array_equal(hd[b], hd[b].T)                         # symmetric to the last bit
        assert np.linalg.eigvalsh(hd[b]).min() > -1e-10               # PSD: w >= 0 and a convex objective
    # unit weights / unit sigma by default; the same pattern as the exact callback
    d0 = eng.hess_gn(Z, X0)["hvals"].cpu().numpy()
    np.testing.assert_allclose(d0[0], prob.gauss_newton_values(Zh[0], X0h[0], None, 1.0), rtol=1e-11, atol=1e-11)
    lam = torch.zeros(B, eng.m, dtype=torch.float64, device="cuda:0")
    exact0 = eng.hess(Z, X0, lam, eng.to_device(sg))["hvals"].cpu().numpy()
    zero_w = eng.hess_gn(Z, X0, torch.zeros(B, H * nx, dtype=torch.float64, device="cuda:0"), eng.to_device(sg))["hvals"]
    np.testing.assert_allclose(zero_w.cpu().numpy(), exact0, rtol=1e-13, atol=1e-13)   # both reduce to sigma * d2f


def test_gauss_newton_hessian_rolling_window_and_fp32():
    """rolling-window model: an entry sums the blocks of every step whose window holds both variables"""
    d, W, b = load_case("roll3_discret_rev")
    from test_gpu_rolling import _engine as rolling_engine
    eng = rolling_engine(d, W, b, torch.float64, "auto")
    Bn = d["Z"].shape[0]
    H, nx = int(d["H"]), int(d["nx"])
    wh = np.random.default_rng(1).uniform(0.2, 1.0, size=(Bn, H * nx))
    out = eng.hess_gn(eng.to_device(d["Z"]), eng.to_device(d["X0"]), eng.to_device(wh))["hvals"].cpu().numpy()
    for i in range(Bn):
        pr = oracle_problem(d, W, b, i)
        np.testing.assert_allclose(out[i], pr.gauss_newton_values(d["Z"][i], d["X0"][i], wh[i], 1.0), rtol=1e-11, atol=1e-11)
    # fp32 handle, configs[2] dims at a small batch
    net = orc.MLP.random(9, [128, 128, 128], 6, seed=0)
    e32 = _engine(net, 30, 6, 3, 4, "rk4", 0.1, dtype=torch.float32)
    Zh, X0h = orc.synthetic_inputs(4, 30, 6, 3, seed=1)
    hv = e32.hess_gn(e32.to_device(Zh), e32.to_device(X0h))["hvals"].to("cpu", torch.float64).numpy()
    p32 = orc.Problem(net, 30, 6, 3, orc.RK4, 0.1)
    ref = np.stack([p32.gauss_newton_values(Zh[i], X0h[i], None, 1.0) for i in range(4)])
    assert np.abs(hv - ref).max() / np.abs(ref).max() < 1e-4


def test_bound_hessian_callbacks_reevaluate_in_place():
    """CallbackEngine.bind_hess: one ctypes call per evaluation, outputs fixed, inputs read at their current contents."""
    nx, nu, H, B = 2, 1, 20, 33
    net = orc.MLP.random(nx + nu, [64, 64], nx, seed=2)
    eng = _engine(net, H, nx, nu, B)
    eng.set_objective(Q=np.eye(nx), R=0.2 * np.eye(nu))
    prob = orc.Problem(net, H, nx, nu, orc.DISCRET, 1.0, Q=np.eye(nx), R=0.2 * np.eye(nu))
    rng = np.random.default_rng(3)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=6)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    lam, sig = eng.to_device(rng.normal(size=(B, eng.m))), eng.to_device(rng.uniform(0.5, 1.5, size=B))
    wgt = eng.to_device(rng.uniform(0.1, 2.0, size=(B, H * nx)))
    call_h, out_h = eng.bind_hess(Z, X0, lam, sig)
    call_g, out_g = eng.bind_hess(Z, X0, wgt, sig, gauss_newton=True)
    Z2h, _ = orc.synthetic_inputs(B, H, nx, nu, seed=7)
    Z.copy_(eng.to_device(Z2h))                          # new iterate in the SAME tensor
    call_h()
    call_g()
    torch.cuda.synchronize()
    hv, gv = out_h["hvals"].cpu().numpy(), out_g["hvals"].cpu().numpy()
    lh, sh, wh = lam.cpu().numpy(), sig.cpu().numpy(), wgt.cpu().numpy()
    for b in (0, 16, B - 1):
        np.testing.assert_allclose(hv[b], prob.hessian_values(Z2h[b], X0h[b], lh[b], sh[b]), rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(gv[b], prob.gauss_newton_values(Z2h[b], X0h[b], wh[b], sh[b]), rtol=1e-11, atol=1e-11)
    with pytest.raises(ValueError, match="sigma"):
        eng.bind_hess(Z, X0, wgt, None, gauss_newton=True)
