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
