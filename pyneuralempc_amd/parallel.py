"""Multi-GPU: independent MPC problem instances are sharded over ranks (one process per GPU);
there is no collective on the callback path.  The only exchange is one all-gather (RCCL over xGMI
with backend "nccl"; "gloo" in CPU tests) of the solved first controls u0 per MPC step --
(B/G)*nu elements per rank, latency-bound.  The reference has no distributed code."""
import torch
import torch.distributed as dist


def shard_bounds(B, rank, world):
    """Contiguous [lo, hi) slice of B problems owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def first_controls(Z, H, nx, nu):
    """u0 of every problem: z[H*nx : H*nx+nu]  (layout optimizer/ipopt.py:20-28)."""
    return Z[:, H * nx:H * nx + nu].contiguous()


def init_u0_comm(engine, group=None):
    """Collective: give `engine`'s handle an RCCL communicator spanning the ranks of `group` (nempc_comm_init).  Rank 0
    draws the ncclUniqueId (nempc_comm_unique_id) and torch.distributed carries its 128 bytes to the other ranks --
    the process group is only the bootstrap channel; the all-gather itself is issued by libnempc.so on RCCL."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("init_u0_comm needs an initialised torch.distributed process group")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_device = dist.get_backend(group) == "nccl"
    # byte 0 = "rank 0 obtained an id": if it could not (RCCL missing), EVERY rank raises instead of the others waiting
    # in the broadcast for an id that never comes
    raw, err = bytes(129), None
    if rank == 0:
        try:
            raw = b"\x01" + type(engine).comm_unique_id()
        except Exception as e:      # noqa: BLE001  (reported below, on every rank)
            err = e
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    if on_device:
        t = t.to(engine.device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    got = bytes(t.cpu().numpy().tobytes())
    if got[0] != 1:
        raise RuntimeError(f"init_u0_comm: rank 0 could not create an RCCL unique id ({err})" if rank == 0 else
                           "init_u0_comm: rank 0 could not create an RCCL unique id")
    engine.comm_init(world, rank, got[1:])
    return engine.comm


def _shard_counts(b, device, total, world, rank, group):
    if total is not None:
        counts = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
        if counts[rank] != b:
            raise ValueError("local shard size does not match shard_bounds(total, rank, world)")
        return counts
    sizes = torch.tensor([b], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    return [int(s.item()) for s in all_sizes]


def allgather_u0(u0_local, total=None, group=None, engine=None, out=None):
    """Gather (b_r, nu) per rank into (sum_r b_r, nu), rank-major.  Always returns a tensor the caller owns: a fresh
    one, or `out` (engine path with equal shards only: (world*b, nu), gathered into in place -- the hot-loop form).

    With `engine` holding a communicator (init_u0_comm) the exchange is libnempc.so's own nempc_allgather_u0 --
    ncclAllGather on RCCL, in place on a persistent buffer; pass `total` (the global problem count) so that no size
    exchange precedes it.  Otherwise it goes through torch.distributed (backend "nccl" = RCCL on a GPU node, "gloo" in
    the CPU tests).  Ragged shards pad to the largest shard."""
    if not (dist.is_available() and dist.is_initialized()):
        return u0_local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, nu = u0_local.shape
    counts = _shard_counts(b, u0_local.device, total, world, rank, group)
    bmax = max(counts)
    equal = all(c == bmax for c in counts)
    if engine is not None and engine.comm is not None:
        if engine.comm != (world, rank):
            raise ValueError("the engine's communicator does not span this process group")
        if out is not None and not equal:
            raise ValueError("allgather_u0: `out` needs equal shards")
        # without `out` the gather lands in the engine's persistent buffer, which the next call (or a reserve /
        # set_box_rows) overwrites: hand back a copy, like the torch.distributed path below does
        res = engine.allgather_u0(u0=u0_local.contiguous(), rows_per_rank=bmax, out=out)
        if equal:
            return res if out is not None else res.clone()
        return torch.cat([res[r * bmax:r * bmax + c] for r, c in enumerate(counts)], dim=0)
    fused = dist.get_backend(group) == "nccl"      # all_gather_into_tensor: a capability of the backend, not a try
    if equal:
        if fused:
            out = torch.empty(world * b, nu, dtype=u0_local.dtype, device=u0_local.device)
            dist.all_gather_into_tensor(out, u0_local.contiguous(), group=group)
            return out
        parts = [torch.empty_like(u0_local) for _ in range(world)]
        dist.all_gather(parts, u0_local.contiguous(), group=group)
        return torch.cat(parts, dim=0)
    padded = torch.zeros(bmax, nu, dtype=u0_local.dtype, device=u0_local.device)
    padded[:b] = u0_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class EvalPipeline:
    """Keep several independent batches in flight on one GPU: `depth` handles, each on its own HIP stream, used in
    rotation.  One evaluation's launch prologue and drain then run under another's full pass -- at C2 dims 35.9 k vs
    33.2 k batch-evals/s for B=1024 and 88 k vs 53 k for B=256 (bench.py `pipelined_two_streams`).  A handle is not
    re-entrant, hence one handle per slot; the batches must be independent (different MPC problem sets).

        pipe = EvalPipeline(lambda: CallbackEngine(W, b, H, nx, nu, max_batch=B), depth=2)
        t0 = pipe.submit(Z0, X00); t1 = pipe.submit(Z1, X01)       # asynchronous
        out0 = pipe.wait(t0)                                         # dict of device tensors owned by slot 0
    """

    def __init__(self, make_engine, depth=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engines = [make_engine() for _ in range(depth)]
        dev = self.engines[0].device
        self.streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        self._next = 0

    def submit(self, Z, X0, want=("f", "grad", "g", "jac_dense")):
        """Launch on the next slot; returns a ticket.  The slot's output tensors are reused by its next submit, so wait
        for (and consume) a ticket before `depth` further submits.  Z / X0 must not be modified in place until the
        ticket has been waited for."""
        i = self._next
        self._next = (i + 1) % len(self.engines)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(st.device))     # inputs produced on the caller's stream
        # the side stream reads Z / X0 after this call returns: tell the caching allocator, so a caller that drops
        # them does not get the memory handed out again under the running kernel.  Overwriting them IN PLACE before
        # wait(ticket) is still a race the caller must avoid (wait first, or submit a copy).
        Z.record_stream(st)
        X0.record_stream(st)
        with torch.cuda.stream(st):
            out = self.engines[i].eval(Z, X0, want)
            ev = torch.cuda.Event()
            ev.record(st)
        return (i, out, ev)

    def wait(self, ticket, stream=None):
        """Make `stream` (default: the current stream) wait for the ticket's evaluation; returns its outputs."""
        _, out, ev = ticket
        (stream or torch.cuda.current_stream(self.streams[0].device)).wait_event(ev)
        return out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
