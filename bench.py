#!/usr/bin/env python3
"""Benchmark of the NMPC callback hot path (BASELINE.json metric:
"NLP callback evals/sec (f + grad f + g + jac g) at batch x H; Jacobian max-abs error vs CPU").

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one batched evaluation of all four callbacks (f, grad f, g, dense jac g) for the B problems
a rank owns, inputs already resident in HBM.  `value` = problem-evals/s over the whole job
(= n_gpus * B * steps / wall; one problem-eval = the four callbacks of ONE NLP instance at one iterate);
`batch_evals_per_s` = value / B per GPU is the north-star reading "evaluations/sec on batch=1024".
Ranks shard independent problems (weak scaling, no data-path collective); the only exchange is one
all-gather of the first controls u0 per MPC step, issued once at the end of the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CONFIGS = {
    # BASELINE.json configs[1] dims (2-state/1-control MLP 2x64, H=20, Euler, fp64) at the north-star batch 1024
    "c2": dict(nx=2, nu=1, hidden=[64, 64], H=20, integrator="discret", DT=1.0, dtype="f64", batch=1024, box=None,
               label="configs[1] dims: 2x1 MLP(2x64), H=20, Euler(Discret), fp64; north_star batch=1024 per GPU"),
    "c2_b256": dict(nx=2, nu=1, hidden=[64, 64], H=20, integrator="discret", DT=1.0, dtype="f64", batch=256, box=None,
                    label="configs[1]: batch=256, 2x1 MLP(2x64), H=20, Euler(Discret), fp64"),
    "c3": dict(nx=6, nu=3, hidden=[128, 128, 128], H=30, integrator="rk4", DT=0.1, dtype="f32", batch=1024, box=None,
               label="configs[2]: batch=1024, 6x3 MLP(3x128), H=30, RK4, fp32"),
    "c5": dict(nx=2, nu=1, hidden=[64, 64], H=50, integrator="discret", DT=1.0, dtype="f64", batch=1024,
               box=(-2.0, 2.0), label="configs[4]: batch=1024, H=50, box state rows, dense jac, fp64"),
}

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F64_TFLOPS = 78.6         # MI355X FP64 vector = matrix peak (spec); v_mfma_f64_16x16x4_f64 at 32 FLOP/clk/SIMD
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 matrix (f32-in MFMA) = vector peak


def algorithmic_work(cfg, B, m, n):
    """flops = H*S*(1+nx)*F, F = 2*sum(in*out); dense-contract bytes per problem-eval."""
    nx, nu, H = cfg["nx"], cfg["nu"], cfg["H"]
    dims = [nx + nu] + cfg["hidden"] + [nx]
    F = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:]))
    S = 4 if cfg["integrator"] == "rk4" else 1
    w = 8 if cfg["dtype"] == "f64" else 4
    flops = H * S * (1 + nx) * F
    dense_bytes = w * ((n + nx) + (1 + n + m + m * n))
    compact_bytes = w * ((n + nx) + 1 + n + m + H * nx * (nx + nu))
    return dict(flops=flops * B, dense_bytes=dense_bytes * B, compact_bytes=compact_bytes * B)


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands a one-GPU job
    a share of the host, and more OpenMP threads than that share only get throttled)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return n


def cpu_baseline(cfg, seconds=12.0):
    """C/OpenMP oracle (oracle/nempc_oracle.c) on the host cores this job may use, bounded sample of the same workload."""
    from oracle import nempc_oracle as orc
    from oracle.c_oracle import COracle
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[cfg["integrator"]]
    net = orc.MLP.random(cfg["nx"] + cfg["nu"], cfg["hidden"], cfg["nx"], seed=0)
    prob = orc.Problem(net, cfg["H"], cfg["nx"], cfg["nu"], kind, cfg["DT"], box=cfg["box"])
    Bs = 512
    Z, X0 = orc.synthetic_inputs(Bs, cfg["H"], cfg["nx"], cfg["nu"], seed=1)
    co = COracle(prob)
    nthr = min(usable_cpus(), 128)
    co.eval(Z, X0, nthreads=nthr)  # warm
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        co.eval(Z, X0, nthreads=nthr)
        reps += 1
    dt = time.perf_counter() - t0
    # single-thread, reference-shaped NumPy loop (per-problem, dense assembly) on a small sample, for context
    t1 = time.perf_counter()
    nref = 0
    while time.perf_counter() - t1 < 3.0:
        prob.objective(Z[nref % Bs]); prob.gradient(Z[nref % Bs])
        prob.constraints(Z[nref % Bs], X0[nref % Bs]); prob.jacobian(Z[nref % Bs], X0[nref % Bs])
        nref += 1
    ref_rate = nref / (time.perf_counter() - t1)
    return {"value": Bs * reps / dt, "unit": "problem-evals/s", "cores": int(co.threads_used), "kind": "port",
            "sample": f"{reps} x {Bs} problem-evals of the same workload in {dt:.1f}s, C/OpenMP oracle "
                      f"(oracle/nempc_oracle.c) on {nthr} threads; host has {os.cpu_count()} logical cpus, "
                      f"{usable_cpus()} usable by this job (affinity / cgroup quota)",
            "numpy_reference_shaped_1core": ref_rate}, prob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (default: the config's)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "valu", "mfma", "mfma_tile"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--hessian", action="store_true", help="also time the Lagrangian-Hessian callback (reported apart)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    # one process per GPU; NEMPC_BENCH_BACKEND=gloo is the one-GPU rehearsal mode (ranks share cuda:0, collectives
    # through gloo) used to exercise this path where only one device exists
    backend = os.environ.get("NEMPC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from oracle import nempc_oracle as orc   # synthetic inputs + checker only (never the thing measured)
    from pyneuralempc_amd import CallbackEngine
    from pyneuralempc_amd.parallel import allgather_u0

    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg["batch"]
    tdtype = torch.float64 if cfg["dtype"] == "f64" else torch.float32
    net = orc.MLP.random(cfg["nx"] + cfg["nu"], cfg["hidden"], cfg["nx"], seed=0)
    eng = CallbackEngine(net.W, net.b, cfg["H"], cfg["nx"], cfg["nu"], integrator=cfg["integrator"], DT=cfg["DT"],
                         dtype=tdtype, device=dev, max_batch=B, kernel=args.kernel)
    if cfg["box"] is not None:
        eng.set_box_rows(*cfg["box"])
    Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1 + rank)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    want = ("f", "grad", "g", "jac_dense")
    u0_off = cfg["H"] * cfg["nx"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step, _outs = eng.bind(Z, X0, want)          # one ctypes call per step: keeps the host out of the way
    for _ in range(args.warmup):
        step()
    if dist is not None:
        allgather_u0(Z[:, u0_off:u0_off + cfg["nu"]].contiguous())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        gathered = allgather_u0(Z[:, u0_off:u0_off + cfg["nu"]].contiguous())
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        assert gathered.shape[0] == world * B

    # ---- per-kernel timing with HIP events on the launch stream (torch's current stream is the one the
    # engine launches on): rows kernel alone, then the post kernels alone
    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / reps

    def per_step_us(fn, reps):
        """one HIP event pair per evaluation -> (p10, median, p90) in microseconds"""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        fn(); torch.cuda.synchronize(dev)
        evs[0].record()
        for i in range(reps):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize(dev)
        d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]) * 1e3
        return [float(np.percentile(d, q)) for q in (10, 50, 90)]

    reps = max(args.steps, 50)
    t_rows = timed(eng.bind(Z, X0, ("g", "jac_tiles"))[0], reps)
    step_pcts = per_step_us(step, reps)
    t_all = timed(step, reps)
    work = algorithmic_work(cfg, B, eng.m, eng.n)
    peak_tf = PEAK_F64_TFLOPS if cfg["dtype"] == "f64" else PEAK_F32_TFLOPS
    ach_tf = work["flops"] / t_rows / 1e12
    t_step = wall / args.steps                      # the timed region itself (max over ranks)
    ach_gbs = work["dense_bytes"] / t_step / 1e9
    ai = work["flops"] / work["dense_bytes"]
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)

    # ---- two independent batches in flight: a second handle on a second HIP stream, evaluations alternating.  The
    # launch prologue / drain of one evaluation overlaps the other's full pass (reported apart; `value` above is the
    # single-stream figure)
    pipe_info = None
    if rank == 0 or dist is None:
        eng2 = CallbackEngine(net.W, net.b, cfg["H"], cfg["nx"], cfg["nu"], integrator=cfg["integrator"], DT=cfg["DT"],
                              dtype=tdtype, device=dev, max_batch=B, kernel=args.kernel)
        if cfg["box"] is not None:
            eng2.set_box_rows(*cfg["box"])
        Zb, X0b = (eng2.to_device(a) for a in orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=101 + rank))
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        with torch.cuda.stream(sa):
            step_a, _ = eng.bind(Z, X0, want)
        with torch.cuda.stream(sb):
            step_b, _ = eng2.bind(Zb, X0b, want)
        torch.cuda.synchronize(dev)
        for k in range(20):
            (step_a if k % 2 == 0 else step_b)()
        torch.cuda.synchronize(dev)
        tp = time.perf_counter()
        for k in range(args.steps):
            (step_a if k % 2 == 0 else step_b)()
        torch.cuda.synchronize(dev)
        tp = (time.perf_counter() - tp) / args.steps
        pipe_info = {"us_per_eval": tp * 1e6, "batch_evals_per_s": 1.0 / tp,
                     "note": "two handles on two HIP streams, independent batches alternating (this rank only)"}
        del eng2
        step, _outs = eng.bind(Z, X0, want)      # rebind the first handle to the default stream

    # ---- batched on-device solver + the one collective of the design: all-gather of the solved u0
    solver_info = None
    # (RK4 Lagrangian blocks on the generic kernel take minutes at C3 dims: only with the matrix-core pipeline)
    if cfg["box"] is None and (cfg["integrator"] != "rk4" or eng.kernel_variant != "valu"):
        lbv = np.concatenate([np.full(cfg["H"] * cfg["nx"], -3.0), np.full(cfg["H"] * cfg["nu"], -0.5)])
        Xs = eng.to_device(np.random.default_rng(100 + rank).uniform(-0.5, 0.5, size=(B, cfg["nx"])))
        # fp32 configs: tolerances an fp32 iterate can reach
        tols = {} if cfg["dtype"] == "f64" else dict(tol_constraint=1e-4, tol_step=1e-4, mu_min=1e-5, reg=1e-6)
        eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=5, **tols)   # warm
        barrier()
        ts = time.perf_counter()
        Zs, st, its = eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=40, **tols)
        torch.cuda.synchronize(dev)
        t_solve = time.perf_counter() - ts
        u0 = Zs[:, u0_off:u0_off + cfg["nu"]].contiguous()
        tg = time.perf_counter()
        allu0 = allgather_u0(u0)
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter() - tg
        conv = torch.tensor([float((st == 0).sum().item()), t_solve], dtype=torch.float64, device=dev)
        if dist is not None:
            c2 = conv.clone()
            dist.all_reduce(c2[0:1], op=dist.ReduceOp.SUM)
            dist.all_reduce(c2[1:2], op=dist.ReduceOp.MAX)
            conv = c2
        solver_info = {"mpc_solves_per_s": world * B / float(conv[1].item()), "iterations": its,
                       "converged_frac": float(conv[0].item()) / (world * B), "solve_ms": float(conv[1].item()) * 1e3,
                       "allgather_u0_us": t_gather * 1e6, "gathered_rows": int(allu0.shape[0]),
                       "note": "SQP + Riccati, exact Lagrangian blocks, bounds |x|<=3 |u|<=0.5 via log barrier, <=40 iterations"}

    hess_info = None
    if args.hessian and (cfg["integrator"] != "rk4" or eng.kernel_variant != "valu"):
        lam = torch.randn(B, eng.m, dtype=tdtype, device=dev)
        sig = torch.ones(B, dtype=tdtype, device=dev)
        t_h = timed(lambda: eng.hess(Z, X0, lam, sig), max(args.steps // 4, 10))
        hess_info = {"hess_us": t_h * 1e6, "nnz_hess": eng.nnz_hess, "hess_batch_evals_per_s": 1.0 / t_h}

    # ---- accuracy vs the CPU oracle on a sample (the metric's second half)
    res = eng.eval(Z[:32].contiguous(), X0[:32].contiguous(), want)
    res = {k: v.to("cpu", torch.float64).numpy() for k, v in res.items()}
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[cfg["integrator"]]
    prob = orc.Problem(net, cfg["H"], cfg["nx"], cfg["nu"], kind, cfg["DT"], box=cfg["box"])
    f, grad, g, jac = prob.eval_batch(Zh[:32], X0h[:32])
    errs = {"f": float(np.abs(res["f"] - f).max()), "grad": float(np.abs(res["grad"] - grad).max()),
            "g": float(np.abs(res["g"] - g).max()), "jac": float(np.abs(res["jac_dense"] - jac).max())}

    if rank == 0:
        out = {
            "metric": "nlp_callback_evals_per_sec (f + grad f + g + dense jac g)",
            "value": world * B * args.steps / wall,
            "unit": "problem-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["label"], "batch_per_gpu": B, "H": cfg["H"], "nx": cfg["nx"], "nu": cfg["nu"],
                       "hidden": cfg["hidden"], "integrator": cfg["integrator"], "n": eng.n, "m": eng.m,
                       "row_kernel": eng.kernel_variant, "parallelism": f"problem-sharded x{world}"},
            "batch_evals_per_s": args.steps / wall,
            "jacobian_max_abs_err_vs_cpu": errs["jac"], "max_abs_err_vs_cpu": errs,
            "roofline": {"bound": "mfma", "kernel": eng.last_row_kernel, "achieved": ach_tf,
                         "peak": peak_tf, "unit": "TFLOP/s", "frac": ach_tf / peak_tf, "traffic": None,
                         "kernel_us": t_rows * 1e6, "flops_per_launch": work["flops"],
                         "arithmetic_intensity_dense": ai, "ridge": ridge},
            "roofline_hbm_whole_eval": {"bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": ach_gbs / PEAK_HBM_GBS, "bytes_per_eval_dense_contract":
                                            work["dense_bytes"], "eval_us": t_step * 1e6, "eval_us_event_loop": t_all * 1e6,
                                        "eval_us_p10_median_p90": step_pcts},
        }
        # HBM bytes of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py)
        pmc_file = os.path.join(REPO, "profiles", "r01c_c2_b1024_pmc.json")
        if args.config == "c2" and B == 1024 and eng.kernel_variant == "mfma" and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            for k, v in pmc.items():
                if k.startswith(str(eng.last_row_kernel)) and "hbm_traffic_bytes" in v:
                    out["roofline"]["traffic"] = v["hbm_traffic_bytes"]
                    out["roofline"]["traffic_note"] = ("FETCH_SIZE*2 + WRITE_SIZE per launch, profiles/r01c_c2_b1024_pmc.json; "
                                                       "WRITE_SIZE of this kernel's 8-byte stores is uncalibrated "
                                                       "(algorithmic: 0.51 MB read, 1.3 MB written)")
        if pipe_info:
            out["pipelined_two_streams"] = pipe_info
        if solver_info:
            out["batched_solver"] = solver_info
        if hess_info:
            out["hessian_callback"] = hess_info
        if world == 1 and not args.no_cpu:
            cb, _ = cpu_baseline(cfg)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
