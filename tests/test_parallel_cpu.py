"""CPU, world_size 2 over gloo: problem sharding + the u0 all-gather (the only collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyneuralempc_amd.parallel import allgather_u0, first_controls, shard_bounds


def test_shard_bounds_cover_everything():
    for B in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_first_controls_layout():
    H, nx, nu, B = 4, 2, 3, 5
    Z = torch.arange(B * H * (nx + nu), dtype=torch.float64).reshape(B, -1)
    u0 = first_controls(Z, H, nx, nu)
    assert u0.shape == (B, nu) and torch.equal(u0[1], Z[1, H * nx:H * nx + nu])


def _worker(rank, world, port, B, nu, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(B * nu, dtype=torch.float64).reshape(B, nu)
        lo, hi = shard_bounds(B, rank, world)
        got = allgather_u0(full[lo:hi].clone(), total=B)
        got2 = allgather_u0(full[lo:hi].clone())       # sizes discovered by exchange
        ok = torch.equal(got, full) and torch.equal(got2, full)
        np.save(os.path.join(out_dir, f"ok_{B}_{rank}.npy"), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])   # equal shards, ragged shards
def test_allgather_u0_world2_gloo(tmp_path, B):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, B, 3, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"ok_{B}_{r}.npy")[0] == 1


def test_allgather_is_identity_without_process_group():
    u = torch.ones(3, 2)
    assert allgather_u0(u) is u


def test_bench_launcher_spawns_ranks_before_touching_the_gpu(tmp_path):
    """`bench.py --gpus N` without WORLD_SIZE is only a launcher: it must start the N ranks with the torchrun
    environment without importing torch (hence without any HIP call) in the parent, and report a failed rank."""
    import subprocess
    import sys
    from conftest import REPO
    probe = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.pop("WORLD_SIZE", None)
import bench
started = []
class FakeProc:
    def __init__(self, cmd, env=None):
        started.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        self.returncode = 3 if env["RANK"] == "1" else 0
    def poll(self):
        return self.returncode
    def kill(self):
        pass
bench.subprocess.Popen = FakeProc
sys.argv = ["bench.py", "--gpus", "2", "--steps", "3"]
try:
    bench.main()
except SystemExit as e:
    code = e.code
assert "torch" not in sys.modules, "the launcher imported torch"
assert len(started) == 2 and [s[1]["RANK"] for s in started] == ["0", "1"]
assert all(s[1]["WORLD_SIZE"] == "2" and s[1]["MASTER_ADDR"] == "127.0.0.1" for s in started)
assert started[0][1]["MASTER_PORT"] == started[1][1]["MASTER_PORT"]
assert started[0][0][-4:] == ["--gpus", "2", "--steps", "3"]
assert code == 3, code
print("launcher-ok")
''' % REPO
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "launcher-ok" in r.stdout, r.stdout + r.stderr


def test_bench_rank_refuses_mismatched_world(tmp_path):
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
