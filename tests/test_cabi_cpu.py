"""CPU: the C-ABI library loads and exports every symbol include/nempc.h declares; no compute calls."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "nempc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nempc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pyneuralempc_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"libnempc.so does not export {s}"
    assert sorted(_lib.EXPORTS) == syms
    assert lib.nempc_abi_version() == _lib.ABI_VERSION


def test_config_struct_matches_header_layout():
    from pyneuralempc_amd import _lib
    # 8 int32 + 8 widths + 5 int32 = 21 int32 (84 B) -> padded to 88, + double = 96, + 8 activation codes = 128
    assert ctypes.sizeof(_lib.NempcConfig) == 192       # (ABI v7: + 8 doubles of per-layer activation parameters)
    assert _lib.NempcConfig.act_param.offset == 128
    assert _lib.NempcConfig.rolling_window.offset == 76
    assert _lib.NempcConfig.DT.offset == 88
    assert _lib.NempcConfig.activations.offset == 96
    # the codes of include/nempc.h
    text = open(os.path.join(REPO, "include", "nempc.h")).read()
    for name, code in _lib.ACTIVATION_IDS.items():
        assert re.search(rf"#define NEMPC_ACT_{name.upper()} {code}\b", text), name


def test_create_validates_before_touching_the_device_and_fails_loudly_without_gpu():
    import torch
    from pyneuralempc_amd import _lib
    lib = _lib.load()
    cfg = _lib.NempcConfig()
    h = ctypes.c_void_p()
    cfg.abi_version = 99
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"abi_version" in lib.nempc_last_error()
    cfg.abi_version = _lib.ABI_VERSION
    cfg.dtype, cfg.integrator, cfg.H, cfg.nx, cfg.nu, cfg.n_layers = 0, 0, 0, 2, 1, 1
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.H = 4
    cfg.widths[0] = 3          # last width != nx
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.widths[0] = 2
    cfg.activations[0] = 14           # no such activation
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"activation" in lib.nempc_last_error()
    cfg.activations[0], cfg.act_param[0] = 5, 0.0          # elu needs alpha > 0
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"alpha" in lib.nempc_last_error()
    cfg.activations[0], cfg.act_param[0] = 6, -0.5         # leaky_relu needs alpha >= 0
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"alpha" in lib.nempc_last_error()
    cfg.activations[0], cfg.act_param[0] = 0, 0.0
    cfg.max_batch = 1
    cfg.integrator, cfg.DT = 2, 0.0   # RK4 without DT
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.integrator, cfg.DT = 0, 1.0
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"rolling_window" in lib.nempc_last_error()
    cfg.rolling_window, cfg.integrator = 2, 2   # the reference has no RK4 for rolling-window models
    assert lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h)) == -5
    cfg.rolling_window = 1
    if not torch.cuda.is_available():
        cfg.integrator = 0
        rc = lib.nempc_create(ctypes.byref(cfg), ctypes.byref(h))
        assert rc == -3 and b"no HIP device" in lib.nempc_last_error()
        assert not h.value
    assert lib.nempc_destroy(None) == 0
    assert lib.nempc_eval(None, 1, None, None, None, None, None, None, None, None, None) == -1


def test_comm_entry_points_validate_without_a_device():
    """the RCCL entry points reject bad arguments before they bind RCCL or touch a device"""
    from pyneuralempc_amd import _lib
    lib = _lib.load()
    assert lib.nempc_comm_unique_id(None) == -1
    assert lib.nempc_comm_init(None, 2, 0, None) == -1
    assert lib.nempc_allgather_u0(None, 1, 1, None, None, None, None) == -1
    assert lib.nempc_comm_destroy(None) == -1
    assert lib.nempc_reserve(None, 4) == -1
    assert _lib.COMM_ID_BYTES == 128


def test_engine_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyneuralempc_amd import CallbackEngine
    import numpy as np
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CallbackEngine([np.zeros((3, 2))], [np.zeros(2)], 4, 2, 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "pyneuralempc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


@pytest.mark.parametrize("num_cus", [32, 64, 256, 304])
def test_launch_planning_covers_every_tile_for_any_cu_count(num_cus):
    """nempc_plan_grid (host arithmetic; the cooperative kernels' tiles-per-workgroup split): sized from the device's
    CU count, never from a literal 256 -- every tile owned exactly once, contiguous runs, sizes differing by at most
    one, at most per_cu workgroups per CU, and the larger runs in front (the first-dispatched workgroups)."""
    from pyneuralempc_amd import _lib
    lib = _lib.load()
    g, q, r = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    for per_cu in (1, 2):
        for ntiles in (1, 2, num_cus - 1, num_cus, num_cus + 1, 2 * num_cus, 2 * num_cus + 1, 1280, 3200, 8191, 130000):
            if ntiles < 1:
                continue
            assert lib.nempc_plan_grid(ntiles, num_cus, per_cu, ctypes.byref(g), ctypes.byref(q), ctypes.byref(r)) == 0
            grid, per, rem = g.value, q.value, r.value
            assert 1 <= grid <= min(ntiles, num_cus * per_cu)
            assert grid == min(ntiles, num_cus * per_cu)          # the chip is filled whenever there is enough work
            begins = [i * per + min(i, rem) for i in range(grid)]
            ends = [b + per + (1 if i < rem else 0) for i, b in enumerate(begins)]
            assert begins[0] == 0 and ends[-1] == ntiles
            assert all(e == b2 for e, b2 in zip(ends[:-1], begins[1:]))
            sizes = [e - b for b, e in zip(begins, ends)]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert lib.nempc_plan_grid(0, num_cus, 1, ctypes.byref(g), ctypes.byref(q), ctypes.byref(r)) == -1
    assert lib.nempc_plan_grid(8, 0, 1, ctypes.byref(g), ctypes.byref(q), ctypes.byref(r)) == -1
