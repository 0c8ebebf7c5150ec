"""GPU: the library's own RCCL entry points (nempc_comm_*, nempc_allgather_u0) on a one-rank communicator -- the only
size a one-GPU box can build (RCCL refuses two ranks on one device) -- plus nempc_reserve and the bound-batch checks.
The N-rank form is exercised by `bench.py --gpus N` on a multi-GPU node and by the gloo tests in
test_parallel_cpu.py for the sharding logic."""
import numpy as np
import pytest
import torch

from oracle import nempc_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(B, dtype=torch.float64, **kw):
    from pyneuralempc_amd import CallbackEngine
    net = orc.MLP.random(3 + kw.get("n_extra", 0), [32, 32], 2, seed=3)
    return CallbackEngine(net.W, net.b, 6, 2, 1, dtype=dtype, device="cuda:0", max_batch=B, **kw), net


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_allgather_u0_single_rank_rccl(dtype):
    from pyneuralempc_amd import CallbackEngine
    B = 37
    eng, _ = _engine(B, dtype)
    assert eng.comm is None
    with pytest.raises(RuntimeError, match="comm_init"):
        eng.allgather_u0(Z=torch.zeros(B, eng.n, dtype=dtype, device="cuda:0"))
    eng.comm_init(1, 0, CallbackEngine.comm_unique_id())
    assert eng.comm == (1, 0)
    Z = torch.randn(B, eng.n, dtype=dtype, device="cuda:0")
    got = eng.allgather_u0(Z=Z)
    torch.cuda.synchronize()
    assert got.shape == (B, 1) and torch.equal(got, Z[:, 12:13])
    # explicit u0, padded slot (ragged shards pad to the largest): pad rows are zero
    u0 = torch.randn(B, 1, dtype=dtype, device="cuda:0")
    got = eng.allgather_u0(u0=u0, rows_per_rank=B + 3)
    torch.cuda.synchronize()
    assert got.shape == (B + 3, 1) and torch.equal(got[:B], u0) and not got[B:].any()
    with pytest.raises(ValueError):
        eng.allgather_u0(Z=Z, u0=u0)
    # growing the workspaces keeps the communicator (nempc_reserve does not re-create the handle)
    eng.reserve(4 * B)
    assert eng.max_batch == 4 * B
    Z2 = torch.randn(4 * B, eng.n, dtype=dtype, device="cuda:0")
    assert torch.equal(eng.allgather_u0(Z=Z2), Z2[:, 12:13])


def test_parallel_helper_uses_the_engine_comm_single_process_group():
    """init_u0_comm + allgather_u0(engine=...) through a one-rank torch.distributed group (nccl = RCCL)"""
    import os
    import socket
    import torch.distributed as dist
    from pyneuralempc_amd.parallel import allgather_u0, init_u0_comm
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        eng, _ = _engine(16)
        assert init_u0_comm(eng) == (1, 0)
        u0 = torch.randn(16, 1, dtype=torch.float64, device="cuda:0")
        assert torch.equal(allgather_u0(u0, total=16, engine=eng), u0)
        assert torch.equal(allgather_u0(u0, total=16), u0)          # torch.distributed path, same answer
    finally:
        dist.destroy_process_group()


def test_reserve_keeps_weights_objective_and_results():
    eng, net = _engine(4)
    eng.set_objective(Q=np.diag([2.0, 3.0]), R=[[0.5]])
    eng.set_box_rows(-1.0, 1.0)
    Zh, X0h = orc.synthetic_inputs(40, 6, 2, 1, seed=5)
    small = eng.eval_numpy(Zh[:4], X0h[:4])
    big = eng.eval_numpy(Zh, X0h)              # grows through nempc_reserve
    assert eng.max_batch == 40
    for k in small:
        assert np.array_equal(small[k], big[k][:4]), k
    prob = orc.Problem(net, 6, 2, 1, orc.DISCRET, Q=np.diag([2.0, 3.0]), R=np.array([[0.5]]), box=(-1.0, 1.0))
    f, grad, g, jac = prob.eval_batch(Zh, X0h)
    np.testing.assert_allclose(big["f"], f, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(big["jac_dense"], jac, rtol=1e-12, atol=1e-12)
    lam = torch.randn(40, eng.m, dtype=torch.float64, device="cuda:0")
    sig = torch.ones(40, dtype=torch.float64, device="cuda:0")
    hv = eng.hess(eng.to_device(Zh), eng.to_device(X0h), lam, sig)["hvals"].cpu().numpy()
    ref = np.stack([prob.hessian_values(Zh[i], X0h[i], lam[i].cpu().numpy(), 1.0) for i in range(40)])
    np.testing.assert_allclose(hv, ref, rtol=1e-11, atol=1e-11)


def test_bound_extras_smaller_than_the_batch_are_refused():
    """a (1,H,ne) extras tensor left bound by a B=1 call must not be read for B problems"""
    from pyneuralempc_amd._lib import NempcError
    eng, _ = _engine(8, n_extra=2)
    Zh, X0h = orc.synthetic_inputs(8, 6, 2, 1, seed=2)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    eng.bind_extra(torch.zeros(1, 6, 2, dtype=torch.float64, device="cuda:0"))
    eng.eval(Z[:1].contiguous(), X0[:1].contiguous())
    with pytest.raises(ValueError, match="bind_extra"):
        eng.eval(Z, X0)
    with pytest.raises(ValueError, match="bind_extra"):
        eng.solve(X0)
    # and the C ABI itself refuses it even when the Python check is bypassed
    import ctypes
    rc = eng.lib.nempc_eval(eng._handle, 8, ctypes.c_void_p(Z.data_ptr()), ctypes.c_void_p(X0.data_ptr()), None, None,
                            ctypes.c_void_p(eng._out("g", (8, eng.m)).data_ptr()), None, None, None, None)
    assert rc == -1 and b"bound extras" in eng.lib.nempc_last_error()
    eng.bind_extra(torch.zeros(8, 6, 2, dtype=torch.float64, device="cuda:0"))
    eng.eval(Z, X0)


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` end to end on the one GPU a box has: the launcher starts two ranks, both share cuda:0 and
    the collectives go through gloo (NEMPC_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device) -- every line of
    the N > 1 code path except the RCCL calls themselves: sharded inputs per rank, the u0 all-gather inside the timed
    loop once per MPC step with the returned shard checked, barrier-bracketed timing with the MAX over ranks, the
    batched solve + gather of the solved u0, rank 0's single JSON line."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NEMPC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import tempfile
    with tempfile.TemporaryDirectory() as cwd:
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "34", "--warmup", "3",
                            "--batch", "64", "--no-cpu", "--no-other-configs", "--no-hessian", "--prime-ms", "0",
                            "--solver-iters", "10"], capture_output=True, text=True, env=env, timeout=600, cwd=cwd)
        assert r.returncode == 0, r.stderr[-2000:]
        details = json.load(open(os.path.join(cwd, "bench_details.json")))      # the detailed record (side file)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0]), r.stdout[-2000:]      # ONE line, the last one
    assert len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 34 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 64 and out["config"]["parallelism"] == "problem-sharded x2"
    # whole-job aggregate: both ranks' problems over the slowest rank's time
    assert abs(out["value"] - 2 * 64 * 34 / (out["ms_per_step"] * 34 * 1e-3)) / out["value"] < 1e-9
    assert details["value"] == out["value"] and details["ms_per_step"] == out["ms_per_step"]
    ag = details["allgather_u0"]
    assert ag["issued_in_timed_loop"] == 2 and ag["rows_gathered"] == 128 and ag["latency_us"] > 0     # 34 steps / 17
    assert out["summary"]["allgather_u0_us"] > 0
    assert details["batched_solver"]["gathered_rows"] == 128
    assert out["jacobian_max_abs_err_vs_cpu"] < 1e-12
    assert out["roofline"]["frac"] > 0 and "FUSE = true" in out["roofline"]["kernel"]


_TWO_RANK_WORKER = r'''
import os, sys, json
rank, world, port, repo = int(sys.argv[1]), 2, sys.argv[2], sys.argv[3]
sys.path.insert(0, repo)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
import numpy as np, torch
import torch.distributed as dist
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
from pyneuralempc_amd.parallel import init_u0_comm
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)          # bootstrap + the reference answer; the gather under test is RCCL
net = orc.MLP.random(3, [32, 32], 2, seed=3)
rows = [37, 22]                                                        # ragged shards: slots pad to the largest
eng = CallbackEngine(net.W, net.b, 6, 2, 1, dtype=torch.float64, device=f"cuda:{rank}", max_batch=64)
init_u0_comm(eng)                                                      # nempc_comm_unique_id on rank 0, broadcast, nempc_comm_init
assert eng.comm == (2, rank)
ok = True
for rep in range(3):
    u0 = torch.full((rows[rank], 1), float(10 * rank + rep), dtype=torch.float64, device=f"cuda:{rank}") + \
        torch.arange(rows[rank], dtype=torch.float64, device=f"cuda:{rank}")[:, None] / 100
    got = eng.allgather_u0(u0=u0, rows_per_rank=max(rows))             # in-place ncclAllGather of the padded slots
    torch.cuda.synchronize()
    parts = [torch.zeros(max(rows), 1, dtype=torch.float64) for _ in range(world)]
    mine = torch.zeros(max(rows), 1, dtype=torch.float64)
    mine[:rows[rank]] = u0.cpu()
    dist.all_gather(parts, mine)                                       # gloo: the answer
    ok = ok and torch.equal(got.cpu(), torch.cat(parts))
Z = torch.randn(rows[0], eng.n, dtype=torch.float64, device=f"cuda:{rank}")
g2 = eng.allgather_u0(Z=Z)
torch.cuda.synchronize()
ok = ok and g2.shape == (2 * rows[0], 1) and torch.equal(g2[rank * rows[0]:(rank + 1) * rows[0]], Z[:, 12:13])
print(json.dumps({"rank": rank, "ok": bool(ok)}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_rccl_allgather_on_two_gpus(tmp_path):
    """nempc_comm_init + nempc_allgather_u0 over RCCL with TWO ranks, ragged shards, against gloo's answer -- on a box that
    has two GPUs (one process per GPU, started before this test touches a second device; skipped on the one-GPU boxes the
    round's own runs get).  The path `bench.py --gpus N` takes for configs[3]."""
    import json
    import os
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_TWO_RANK_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, repo], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {se[-2000:]}"
        line = [ln for ln in so.splitlines() if ln.startswith("{")][-1]
        assert json.loads(line) == {"rank": r, "ok": True}
