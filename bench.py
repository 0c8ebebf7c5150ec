#!/usr/bin/env python3
"""Benchmark of the NMPC callback hot path (BASELINE.json metric:
"NLP callback evals/sec (f + grad f + g + jac g) at batch x H; Jacobian max-abs error vs CPU").

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...            # spawns N ranks itself (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one batched evaluation of all four callbacks (f, grad f, g, dense jac g) for the B problems
a rank owns, inputs already resident in HBM.  `value` = problem-evals/s over the whole job
(= n_gpus * B * steps / wall; one problem-eval = the four callbacks of ONE NLP instance at one iterate);
`batch_evals_per_s` = value / B per GPU is the north-star reading "evaluations/sec on batch=1024".
Ranks shard independent problems (weak scaling, no data-path collective).  The only exchange is the all-gather of
the first controls u0, once per MPC step: with N > 1 it is issued INSIDE the timed loop, through libnempc.so's own
RCCL call (nempc_allgather_u0), once every `--evals-per-mpc-step` evaluations (an MPC step = one NLP solve = that
many callback evaluations; default 17, the median number of iterations the batched solver needs on this workload --
`batched_solver.iters_to_converge_p50` -- each of which evaluates the callbacks at least once).

Timing: W warm-up steps; a barrier and a device synchronize, exactly K steps, a device synchronize that stops the rank's
clock, a barrier, and the MAX over ranks.  That region is run twice: once as the process finds the GPU
(`value_from_cold_gpu`), and once after the device has been brought to its sustained clock by `--prime-ms` (default 40)
of the same step, untimed and reported (`clock_priming`) -- `value`.  `--prime-ms 0` runs the cold region only.

With `--gpus N > 1` and no WORLD_SIZE in the environment this process only launches the N ranks (before anything
touches the GPU) and relays rank 0's JSON line; a failed rank makes the exit code non-zero.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

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
    # configs[3]: the C3 problem at a GLOBAL batch of 4096 sharded over the ranks (512 per GPU on 8 GPUs)
    "c4": dict(nx=6, nu=3, hidden=[128, 128, 128], H=30, integrator="rk4", DT=0.1, dtype="f32", batch=None,
               global_batch=4096, box=None,
               label="configs[3]: batch=4096 sharded over the GPUs, 6x3 MLP(3x128), H=30, RK4, fp32, all-gather of u0"),
    "c5": dict(nx=2, nu=1, hidden=[64, 64], H=50, integrator="discret", DT=1.0, dtype="f64", batch=1024,
               box=(-2.0, 2.0), label="configs[4]: batch=1024, H=50, box state rows, dense jac, fp64"),
}

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F64_TFLOPS = 78.6         # MI355X FP64 vector = matrix peak (spec); v_mfma_f64_16x16x4_f64 at 32 FLOP/clk/SIMD
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 matrix (f32-in MFMA) = vector peak
MALL_BYTES = 256 << 20         # Infinity Cache: an output that fits is not an HBM stream when rewritten in place


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


# ------------------------------------------------------------------------------------------------------------------
# the line the driver reads
# ------------------------------------------------------------------------------------------------------------------
DRIVER_LINE_MAX = 4096         # bytes; tests/test_bench_line_cpu.py holds the line to it
DETAILS_FILE = "bench_details.json"


def _r(x, sig=6):
    """a float at `sig` significant digits (the line is for a parser and a reader, not for bit-exact replay)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def _short_kernel(name):
    """kernel name without the prose that follows it in the detailed record"""
    name = str(name)
    for stop in (":", " (", " writing"):
        i = name.find(stop)
        if i > 0:
            name = name[:i]
    return name[:64]


def _compact_roofline(rf):
    out = _pick(rf, "bound", "achieved", "peak", "unit", "frac", "traffic")
    out["kernel"] = _short_kernel(rf.get("kernel", ""))
    for k in ("kernel_us", "eval_us"):
        if k in rf:
            out["us"] = _r(rf[k])
            break
    return out


def _leg(us, rf):
    """one leg of the summary: microseconds, roofline fraction and which roofline"""
    return {"us": _r(us, 5), "frac": _r(rf.get("frac"), 4), "bound": rf.get("bound")}


def driver_line(full):
    """The ONE stdout line of the driver's contract, from the detailed result `full`: the contract's keys, `roofline`,
    `cpu_baseline` and one compact `summary` of the other legs (microseconds and roofline fraction each, no prose).
    Everything else stays in DETAILS_FILE.  Strict JSON (no NaN / Infinity tokens), < DRIVER_LINE_MAX bytes."""
    # (the contract's own numbers at full precision: value == n_gpus * B * steps / (ms_per_step * steps) has to hold to the bit)
    line = {k: _sanitize(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                             "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in full}
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, "workload", "batch_per_gpu", "H", "nx", "nu", "hidden", "integrator", "n", "m",
                           "row_kernel", "parallelism", "clock")
    for k in ("batch_evals_per_s", "jacobian_max_abs_err_vs_cpu", "value_from_cold_gpu"):
        if full.get(k) is not None:
            line[k] = _r(full[k])
    if "roofline" in full:
        line["roofline"] = _compact_roofline(full["roofline"])
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind")
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:100]
    summ = {}
    for key, val in full.items():
        if key.startswith("roofline_") and isinstance(val, dict) and "frac" in val:
            summ[key[len("roofline_"):]] = _leg(val.get("kernel_us", val.get("eval_us")), val)
    if "steady_state" in full:
        summ["steady_state"] = {k: _r(v, 4) for k, v in full["steady_state"].items() if not isinstance(v, str)}
    hc = full.get("hessian_callback")
    if hc:
        summ["hess_exact"] = _leg(hc["exact"]["us"], hc["exact"]["roofline"])
        summ["hess_gn"] = _leg(hc["gauss_newton"]["us"], hc["gauss_newton"]["roofline"])
    if "sparse_contract" in full:
        summ["sparse"] = _leg(full["sparse_contract"]["us"], full["sparse_contract"]["roofline"])
    bs = full.get("batched_solver")
    if bs:
        summ["solver"] = _pick(bs, "mpc_solved_per_s", "iterations", "converged_frac", "solve_ms")
    ag = full.get("allgather_u0")
    if ag:
        summ["allgather_u0_us"] = _r(ag.get("latency_us"), 4)
    for nm, e in (full.get("other_configs") or {}).items():
        o = {"dense": _leg(e["ms_per_step"] * 1e3, e["roofline"]), "jac_err": _r(e["max_abs_err_vs_cpu"]["jac"], 3)}
        if "ms_per_step_rotating_outputs" in e:
            o["dense"]["us_rotating"] = _r(e["ms_per_step_rotating_outputs"] * 1e3, 5)
        if "sparse_contract" in e:
            o["sparse"] = _leg(e["sparse_contract"]["us"], e["sparse_contract"]["roofline"])
        if "hessian_callback" in e:
            o["hess_exact"] = _leg(e["hessian_callback"]["exact"]["us"], e["hessian_callback"]["exact"]["roofline"])
        if "batched_solver" in e:
            o["solver"] = _pick(e["batched_solver"], "mpc_solved_per_s", "iterations", "converged_frac")
        summ[nm] = o
    for nm, e in (full.get("layered_path") or {}).items():
        summ["layered_" + nm] = {"eval_us": _r(e["evaluation"]["us"], 5), "eval_frac": _r(e["evaluation"]["frac_of_matrix_peak"], 4),
                                 "hess_us": _r(e["hessian_callback"]["us"], 5),
                                 "hess_frac": _r(e["hessian_callback"]["frac_of_matrix_peak"], 4)}
    for nm, e in (full.get("narrow_networks") or {}).items():
        summ["narrow_" + nm] = {"eval_us": _r(e["evaluation"]["us"], 5), "eval_frac": _r(e["evaluation"]["frac_of_matrix_peak"], 4),
                                "variant": e.get("kernel_variant")}
    if "shard_c4" in full:
        summ["shard_c4_b512"] = _pick(full["shard_c4"], "us", "frac", "problem_evals_per_s")
    if "solver_c3" in full:
        summ["solver_c3"] = {k[len("budget_"):]: _pick(v, "converged_frac", "solve_ms") for k, v in full["solver_c3"].items()
                             if k.startswith("budget_")}
    if summ:
        line["summary"] = summ
    line["details"] = DETAILS_FILE
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= DRIVER_LINE_MAX:          # never hand the driver an oversized line: drop the summary, keep the contract
        line["summary"] = {"dropped": "summary exceeded the line budget; see " + DETAILS_FILE}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return text


def _sanitize(x):
    """NaN / Infinity -> None, so that the side file is strict JSON, too"""
    if isinstance(x, float):
        return x if (x == x and x not in (float("inf"), float("-inf"))) else None
    if isinstance(x, dict):
        return {str(k): _sanitize(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sanitize(v) for v in x]
    return x


def write_details(full, also_stderr=True, name=None):
    """the detailed record: DETAILS_FILE next to bench.py's caller (and under gpurun_out/ when that directory exists, so
    that a gpurun call brings it back), echoed on stderr"""
    text = json.dumps(_sanitize(full), allow_nan=False, indent=1)
    name = name or DETAILS_FILE
    paths = [os.path.join(os.getcwd(), name)]
    if os.path.isdir(os.path.join(REPO, "gpurun_out")):
        paths.append(os.path.join(REPO, "gpurun_out", os.path.basename(name)))
    for pth in paths:
        try:
            with open(pth, "w") as fh:
                fh.write(text + "\n")
        except OSError:
            pass
    if also_stderr:
        sys.stderr.write("bench.py details: " + json.dumps(_sanitize(full), allow_nan=False) + "\n")
        sys.stderr.flush()


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


def oracle_problem(cfg):
    from oracle import nempc_oracle as orc
    kind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[cfg["integrator"]]
    net = orc.MLP.random(cfg["nx"] + cfg["nu"], cfg["hidden"], cfg["nx"], seed=0)
    return net, orc.Problem(net, cfg["H"], cfg["nx"], cfg["nu"], kind, cfg["DT"], box=cfg["box"])


def cpu_baseline(cfg, seconds=12.0):
    """C/OpenMP oracle (oracle/nempc_oracle.c) on the host cores this job may use, bounded sample of the same workload."""
    from oracle import nempc_oracle as orc
    from oracle.c_oracle import COracle
    _, prob = oracle_problem(cfg)
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
            "numpy_reference_shaped_1core": ref_rate}


# ------------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without a torchrun environment starts the N ranks itself
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n_gpus, argv):
    """Start one child per GPU with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), wait for all,
    return the worst exit code.  Runs BEFORE this process has made any HIP / torch.cuda call: a process that has
    touched the GPU must neither fork workers that use it nor be replaced by another program."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    deadline = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes) and deadline is None:
            deadline = time.time() + 30.0          # a rank died: give the others a moment, then stop them
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            worst = worst or (p.returncode if p.returncode > 0 else 1)
    return worst


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        import numpy as np
        import torch
        self.np, self.torch = np, torch
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != self.world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; launch one rank per GPU "
                             "(or drop WORLD_SIZE and let bench.py start the ranks)")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
        # one process per GPU; NEMPC_BENCH_BACKEND=gloo is the one-GPU rehearsal mode (ranks share cuda:0, collectives
        # through gloo) used to exercise this path where only one device exists
        self.backend = os.environ.get("NEMPC_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if self.backend == "nccl" and self.world > ndev:
            raise SystemExit(f"bench.py: {self.world} ranks but {ndev} visible GPU(s); RCCL needs one GPU per rank "
                             "(NEMPC_BENCH_BACKEND=gloo rehearses the rank logic on fewer)")
        dev_index = local_rank if self.backend == "nccl" else local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.dist = None
        # NEMPC_BENCH_FORCE_DIST=1: bring the process group (and with it the RCCL paths) up even for one rank -- the only
        # way to execute the N > 1 code on a one-GPU box
        if self.world > 1 or os.environ.get("NEMPC_BENCH_FORCE_DIST"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    # -------------------------------------------------------------------------------------------------------------
    def make_engine(self, cfg, B, kernel="auto"):
        from pyneuralempc_amd import CallbackEngine
        torch = self.torch
        net, _ = oracle_problem(cfg)                      # weights only: the oracle's generator defines the workload
        tdtype = torch.float64 if cfg["dtype"] == "f64" else torch.float32
        eng = CallbackEngine(net.W, net.b, cfg["H"], cfg["nx"], cfg["nu"], integrator=cfg["integrator"], DT=cfg["DT"],
                             dtype=tdtype, device=self.dev, max_batch=B, kernel=kernel)
        if cfg["box"] is not None:
            eng.set_box_rows(*cfg["box"])
        return eng

    def prime(self, fn, ms=None):
        """untimed run of `fn` for --prime-ms (at least one call), device drained afterwards: every timing of this file
        starts from the sustained clock, not from whatever the idle time before it left"""
        ms = self.args.prime_ms if ms is None else ms
        tp = time.perf_counter()
        fn()
        while (time.perf_counter() - tp) * 1e3 < ms:
            for _ in range(16):
                fn()
            self.torch.cuda.synchronize(self.dev)
        self.torch.cuda.synchronize(self.dev)
        return (time.perf_counter() - tp) * 1e3

    def timed_events(self, fn, reps, prime_ms=None):
        """average seconds per call between two HIP events on the launch stream (torch's current stream is the one the
        engine launches on).  prime_ms=0 for a `fn` that contains a collective: the priming loop is time-based, and ranks
        must issue the same number of collectives."""
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.prime(fn, prime_ms)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(self.dev)
        return e0.elapsed_time(e1) * 1e-3 / reps

    def per_step_us(self, fn, reps):
        """one HIP event pair per evaluation -> (p10, median, p90) in microseconds"""
        torch, np = self.torch, self.np
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        self.prime(fn)
        evs[0].record()
        for i in range(reps):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize(self.dev)
        d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]) * 1e3
        return [float(np.percentile(d, q)) for q in (10, 50, 90)]

    def check_against_oracle(self, cfg, outs, Zh, X0h):
        """Max abs error of the outputs the TIMED launch left behind against the CPU oracle on three 16-problem slices
        (first, middle, last) of the full batch -- no extra launch, the numbers checked are the numbers timed."""
        np, torch = self.np, self.torch
        _, prob = oracle_problem(cfg)
        B = Zh.shape[0]
        k = min(16, B)
        starts = sorted({0, max(0, B // 2 - k // 2), B - k})
        errs = {"f": 0.0, "grad": 0.0, "g": 0.0, "jac": 0.0}
        scale = {"f": 0.0, "grad": 0.0, "g": 0.0, "jac": 0.0}
        for s0 in starts:
            sl = slice(s0, s0 + k)
            f, grad, g, jac = prob.eval_batch(Zh[sl], X0h[sl])
            for key, ref, name in (("f", f, "f"), ("grad", grad, "grad"), ("g", g, "g"), ("jac", jac, "jac_dense")):
                got = outs[name][sl].to("cpu", torch.float64).numpy()
                errs[key] = max(errs[key], float(np.abs(got - ref).max()))
                scale[key] = max(scale[key], float(np.abs(ref).max()))
        return errs, scale, [int(s) for s in starts]

    # -------------------------------------------------------------------------------------------------------------
    def run_config(self, name, steps, warmup, headline, kernel="auto"):
        """Time one configuration on this rank.  headline=True: the barrier-bracketed wall-clock loop over all ranks
        (the driver's contract) -> `value`; False: HIP-event timing only (the other configs of the default run)."""
        np, torch = self.np, self.torch
        from oracle import nempc_oracle as orc   # synthetic inputs + checker only (never the thing measured)
        args = self.args
        cfg = dict(CONFIGS[name])
        if cfg.get("global_batch"):
            if cfg["global_batch"] % self.world:
                raise SystemExit(f"config {name}: global batch {cfg['global_batch']} does not divide over {self.world} ranks")
            cfg["batch"] = cfg["global_batch"] // self.world
        B = (args.batch if headline and args.batch else 0) or cfg["batch"]
        eng = self.make_engine(cfg, B, kernel)
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1 + self.rank)
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        want = ("f", "grad", "g", "jac_dense")
        H, nx, nu = cfg["H"], cfg["nx"], cfg["nu"]
        work = algorithmic_work(cfg, B, eng.m, eng.n)
        w = 8 if cfg["dtype"] == "f64" else 4
        jac_bytes = B * eng.m * eng.n * w

        step, outs = eng.bind(Z, X0, want)           # one ctypes call per step: keeps the host out of the way
        gather = None
        gather_every = max(1, args.evals_per_mpc_step)
        # a communicator only where a gather follows: the headline loop and the sharded configs[3] solve
        if self.dist is not None and (headline or name == "c4"):
            comm_ok = False
            if self.backend == "nccl":
                # the library's own RCCL call (nempc_comm_init / nempc_allgather_u0).  Should the communicator not come
                # up on some node, the run continues on torch.distributed's all_gather -- the same RCCL underneath -- and
                # the line says so (`allgather_u0.path`, `allgather_u0.cabi_error`); every rank must take the same path.
                from pyneuralempc_amd.parallel import init_u0_comm
                try:
                    init_u0_comm(eng)                 # RCCL communicator owned by the handle
                    ok = 1.0
                except Exception as e:                # noqa: BLE001
                    self.comm_error = f"{type(e).__name__}: {e}"
                    ok = 0.0
                comm_ok = self.sum_over_ranks(ok) == float(self.world)
                if not comm_ok and eng.comm is not None:
                    eng.lib.nempc_comm_destroy(eng._handle)
                    eng._comm = None
            if comm_ok:
                # the exchange is issued on the launch stream, between two evaluations.  On a stream of its own it could run
                # under the next evaluations (it only reads Z) -- measured, that costs MORE: the cross-stream event pair, a
                # second stream to drain and a co-running kernel that takes a workgroup slot from an evaluation sized to fill
                # the chip exactly (one rank, 20-step region: 22.6 us per step against 19.5 serialised; 19.3 against 18.6 over
                # 200 steps).  NEMPC_BENCH_GATHER_STREAM=own keeps the other arrangement for A/B.
                comm_stream = (torch.cuda.Stream(self.dev) if os.environ.get("NEMPC_BENCH_GATHER_STREAM") == "own"
                               else torch.cuda.current_stream(self.dev))
                torch.cuda.synchronize(self.dev)
                launch_gather, gathered_buf = eng.bind_allgather_u0(Z, stream=comm_stream)
                main_stream = torch.cuda.current_stream(self.dev)

                def gather():
                    if comm_stream != main_stream:
                        comm_stream.wait_stream(main_stream)
                    launch_gather()
                    return gathered_buf
            else:
                from pyneuralempc_amd.parallel import allgather_u0, first_controls
                gather = lambda: allgather_u0(first_controls(Z, H, nx, nu), total=self.world * B)   # noqa: E731
        res = {"cfg": cfg, "B": B, "eng": eng}

        if headline:
            # A GPU that has been idle (process start-up, allocation, the barrier) runs its first milliseconds below its
            # sustained clock: the same launch takes 19.9 us in a cold 20-step region and 17.3 us after ~30 ms of load
            # (tools/host_launch_cost.py).  The metric is a sustained rate, so the device is brought to it first --
            # `--prime-ms` of this same step, untimed, reported in the line -- then the W warm-up steps, then the K timed ones.
            def timed_region(with_events=False):
                """W warm-up steps, then exactly `steps` steps between two (barrier + device synchronize) brackets; the
                closing synchronize drains the stream and stops this rank's clock, the closing barrier and the MAX over
                ranks follow -- the slowest rank sets the time, the latency of the barrier collective itself (0.1-0.3 ms,
                as long as a short timed region) does not"""
                for i in range(warmup):
                    step()
                    if gather and (i + 1) % gather_every == 0:
                        gather()
                if gather:
                    gather()
                self.barrier()
                torch.cuda.synchronize(self.dev)
                # HIP events on the launch stream around the K launches: in a REPETITION of the region, not in the one the
                # wall clock brackets -- an event record is a marker packet plus host work, and the pair cost the driver's
                # 20-step region 1.8 us per step (measured: 18.5 against 16.65 us per step), 11 % of what is being timed
                use_ev = with_events
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                if use_ev:
                    ev0.record()                   # HIP events on the launch stream, around the K timed launches
                ng, last = 0, None
                for i in range(steps):
                    step()
                    if gather and (i + 1) % gather_every == 0:
                        last = gather()
                        ng += 1
                if use_ev:
                    ev1.record()
                torch.cuda.synchronize(self.dev)
                w_ = time.perf_counter() - t0
                if use_ev:
                    self.region_event_s = ev0.elapsed_time(ev1) * 1e-3 / steps
                self.barrier()
                return self.max_over_ranks(w_), ng, last

            # the same region twice: as the process finds the GPU (cold, reported next to the headline), then -- the metric
            # is a sustained rate -- after the priming run
            res["wall_cold"] = timed_region()[0] if args.prime_ms > 0 else None
            res["primed_ms"] = self.prime(step) if args.prime_ms > 0 else 0.0
            wall, n_gather, gathered = timed_region()
            res["wall"] = wall
            self.region_event_s = None
            timed_region(with_events=True)                    # the same region once more, bracketed by HIP events
            res["t_region_events"] = self.region_event_s      # average launch-to-launch time of the timed launches
            res["n_gather"] = n_gather
            if gather and n_gather:
                assert gathered.shape[0] == self.world * B
                mine = gathered[self.rank * B:(self.rank + 1) * B]
                assert torch.equal(mine, Z[:, H * nx:H * nx + nu]), "all-gather returned a wrong shard"
        else:
            for _ in range(warmup):
                step()
            torch.cuda.synchronize(self.dev)

        # accuracy of the launch just timed (the metric's second half)
        errs, scale, starts = self.check_against_oracle(cfg, outs, Zh, X0h)
        res["errs"], res["err_scale"], res["checked_slices"] = errs, scale, starts

        reps = max(steps, 50)
        t_rows = self.timed_events(eng.bind(Z, X0, ("g", "jac_tiles"))[0], reps)
        row_kernel = eng.last_row_kernel
        step, outs = eng.bind(Z, X0, want)
        t_all = self.timed_events(step, reps)
        res["dense_from_row_launch"] = eng.last_row_kernel == "rows_coop_kernel+dense"
        res.update(t_rows=t_rows, t_all=t_all, row_kernel=row_kernel, work=work)
        res["step_pcts"] = self.per_step_us(step, reps) if headline else None

        # HBM-bound configs: the dense Jacobian rewritten in place stays in the 256 MB Infinity Cache when it fits.
        # Rotate over enough output buffers that every store stream is larger than the cache -> an HBM rate.
        res["t_all_rotating"] = None
        if jac_bytes * 3 >= MALL_BYTES and jac_bytes >= (32 << 20):
            nbuf = max(2, -(-2 * MALL_BYTES // jac_bytes))       # ring >= 2x the cache
            ring = []
            for _ in range(nbuf):
                ring.append(eng.bind(Z, X0, want, out={"jac_dense": torch.empty_like(outs["jac_dense"])})[0])
            k = [0]

            def rot():
                ring[k[0] % nbuf]()
                k[0] += 1
            res["t_all_rotating"] = self.timed_events(rot, reps)
            res["rotation"] = {"buffers": nbuf, "bytes_each": jac_bytes}
            del ring
        return res

    def roofline_of(self, res):
        """Rooflines of a configuration.  The matrix-core one is computed for the kernel the TIMED loop launches: on the
        compiled shapes that is ONE launch -- rows_coopfx_kernel<..., FUSE = true> (rows + dense Jacobian + objective) --
        so its duration is the launch-to-launch time of the timed launches (HIP events on the launch stream around the
        timed region when there is one, else around a loop of the same step); the row work alone (the unfused
        <..., false> instantiation, which the timed loop never launches) is reported next to it.  Elsewhere the timed step
        is the row kernel plus the assembly launch and the row kernel (timed on its own) is the dominant one.  Returns
        (primary, secondary, extra): primary is the binding roofline -- HBM over the whole evaluation when the
        dense-contract bytes sit below the ridge (C5), the matrix cores otherwise."""
        cfg, work = res["cfg"], res["work"]
        peak_tf = PEAK_F64_TFLOPS if cfg["dtype"] == "f64" else PEAK_F32_TFLOPS
        ai = work["flops"] / work["dense_bytes"]
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        t_eval = res["t_all_rotating"] or res["t_all"]
        fused = res["row_kernel"] == "rows_coopfx_kernel"
        extra = {}
        if fused:
            t_k = res.get("t_region_events") or res["t_all"]
            mfma = {"bound": "mfma", "kernel": "rows_coopfx_kernel<..., FUSE = true>: the launch of the timed loop "
                                               "(rows + dense Jacobian rows + objective in one launch)",
                    "achieved": work["flops"] / t_k / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": work["flops"] / t_k / 1e12 / peak_tf, "traffic": None, "kernel_us": t_k * 1e6,
                    "kernel_us_measured": ("HIP events on the launch stream around the K launches of a repetition of the timed "
                                           "region (same warm-up, same K; an event pair inside the wall-clocked region itself "
                                           "costs it 1.8 us per step at K = 20), / K"
                                           if res.get("t_region_events") else "HIP events around a loop of the timed step"),
                    "kernel_us_event_loop": res["t_all"] * 1e6, "flops_per_launch": work["flops"],
                    "arithmetic_intensity_dense": ai, "ridge": ridge}
            extra["roofline_row_kernel_unfused"] = {
                "bound": "mfma", "kernel": "rows_coopfx_kernel<..., FUSE = false> (g + compact tiles only; NOT the timed launch)",
                "achieved": work["flops"] / res["t_rows"] / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                "frac": work["flops"] / res["t_rows"] / 1e12 / peak_tf, "traffic": None, "kernel_us": res["t_rows"] * 1e6}
        elif res.get("dense_from_row_launch"):
            # the timed step = the cooperative row kernel writing the dense rows itself + the small objective launch: the
            # roofline is taken over the whole step (an upper bound of the row launch's duration); the same kernel without
            # the dense rows (the compact contract) next to it
            mfma = {"bound": "mfma", "kernel": "rows_coop_kernel writing the dense Jacobian rows (the dominant launch of the "
                                               "timed step; objective_kernel follows it, inside kernel_us)",
                    "achieved": work["flops"] / res["t_all"] / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": work["flops"] / res["t_all"] / 1e12 / peak_tf, "traffic": None,
                    "kernel_us": res["t_all"] * 1e6, "flops_per_launch": work["flops"],
                    "frac_compact_contract": work["flops"] / res["t_rows"] / 1e12 / peak_tf,
                    "kernel_us_compact_contract": res["t_rows"] * 1e6,
                    "arithmetic_intensity_dense": ai, "ridge": ridge}
        else:
            mfma = {"bound": "mfma", "kernel": str(res["row_kernel"]) + " (the dominant launch of the timed step; the "
                                               "assembly launch follows it)",
                    "achieved": work["flops"] / res["t_rows"] / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": work["flops"] / res["t_rows"] / 1e12 / peak_tf, "traffic": None,
                    "kernel_us": res["t_rows"] * 1e6, "flops_per_launch": work["flops"],
                    "frac_over_whole_eval": work["flops"] / res["t_all"] / 1e12 / peak_tf, "eval_us": res["t_all"] * 1e6,
                    "arithmetic_intensity_dense": ai, "ridge": ridge}
        hbm = {"bound": "hbm", "kernel": ("whole evaluation: one launch, rows_coopfx_kernel<..., FUSE = true>" if fused else
                                           "whole evaluation (row kernel + post_flat_kernel)"),
               "achieved": work["dense_bytes"] / t_eval / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": work["dense_bytes"] / t_eval / 1e9 / PEAK_HBM_GBS, "traffic": None,
               "bytes_per_eval_dense_contract": work["dense_bytes"], "eval_us": t_eval * 1e6,
               "eval_us_in_place": res["t_all"] * 1e6,
               "note": ("dense Jacobian written to a ring of buffers larger than the 256 MB Infinity Cache"
                        if res["t_all_rotating"] else "outputs rewritten in place (cache-resident when they fit 256 MB)")}
        return ((hbm, mfma) if ai < ridge else (mfma, hbm)) + (extra,)

    # -------------------------------------------------------------------------------------------------------------
    def solver_leg(self, res):
        """batched on-device solver + the all-gather of the SOLVED u0; throughput counts converged
        problems only"""
        np, torch = self.np, self.torch
        cfg, B, eng = res["cfg"], res["B"], res["eng"]
        H, nx, nu = cfg["H"], cfg["nx"], cfg["nu"]
        lbv = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
        Xs = eng.to_device(np.random.default_rng(100 + self.rank).uniform(-0.5, 0.5, size=(B, nx)))
        eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=5)   # warm
        self.barrier()
        ts = time.perf_counter()
        Zs, st, its, per = eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=self.args.solver_iters, return_iterations=True)
        torch.cuda.synchronize(self.dev)
        t_solve = time.perf_counter() - ts
        done = per[st == 0].to("cpu").numpy()
        # iterations after which 95 % of the batch had converged (None: fewer than 95 % did within the budget)
        it95 = int(np.sort(done)[int(np.ceil(0.95 * B)) - 1]) if len(done) >= int(np.ceil(0.95 * B)) else None
        # the same batch at other iteration budgets (this rank only): convergence is a function of the budget
        budgets = {}
        for mi in (40, 160):
            if mi == self.args.solver_iters:
                continue
            torch.cuda.synchronize(self.dev)
            tb = time.perf_counter()
            _, stb, _ = eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=mi)
            torch.cuda.synchronize(self.dev)
            tb = time.perf_counter() - tb
            okb = int((stb == 0).sum().item())
            budgets[str(mi)] = {"solve_ms": tb * 1e3, "converged_frac": okb / B, "mpc_solved_per_s": okb / tb}
        # ... and the budget at which 99 % of the batch is through (from a long solve's per-problem iteration counts), timed
        p99 = None
        if self.world == 1:
            _, stl, _, perl = eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=400, return_iterations=True)
            dl = np.sort(perl[stl == 0].to("cpu").numpy())
            k99 = int(np.ceil(0.99 * B))
            if len(dl) >= k99:
                it99 = int(dl[k99 - 1])
                best = None
                for _ in range(3):
                    torch.cuda.synchronize(self.dev)
                    tb = time.perf_counter()
                    _, stb, _ = eng.solve(Xs, lb=lbv, ub=-lbv, max_iter=it99)
                    torch.cuda.synchronize(self.dev)
                    tb = time.perf_counter() - tb
                    best = tb if best is None else min(best, tb)
                p99 = {"iterations": it99, "solve_ms": best * 1e3, "converged_frac": int((stb == 0).sum().item()) / B,
                       "converged_after_400_iterations": len(dl) / B}
        tg = time.perf_counter()
        if eng.comm is not None:
            allu0 = eng.allgather_u0(Z=Zs)
        else:
            from pyneuralempc_amd.parallel import allgather_u0, first_controls
            allu0 = allgather_u0(first_controls(Zs, H, nx, nu), total=self.world * B)
        torch.cuda.synchronize(self.dev)
        t_gather = time.perf_counter() - tg
        n_ok = self.sum_over_ranks(float((st == 0).sum().item()))
        t_solve = self.max_over_ranks(t_solve)
        total = self.world * B
        return {"mpc_solved_per_s": n_ok / t_solve, "problems_per_s_incl_unconverged": total / t_solve,
                "iterations": its, "converged_frac": n_ok / total, "iters_to_95pct": it95,
                "iters_to_converge_p50": (float(np.median(done)) if len(done) else None), "solve_ms": t_solve * 1e3,
                "allgather_u0_us": t_gather * 1e6, "gathered_rows": int(allu0.shape[0]),
                "allgather_path": "nempc_allgather_u0 (RCCL)" if eng.comm is not None else
                                  ("torch.distributed/" + self.backend if self.dist is not None else "single rank"),
                "other_budgets_this_rank": budgets, "budget_for_99pct_converged": p99,
                "note": "SQP + Riccati, exact Lagrangian blocks, bounds |x|<=3 |u|<=0.5 "
                        + (f"and the handle's box rows (states in [{cfg['box'][0]}, {cfg['box'][1]}], taken as state bounds) "
                           if cfg["box"] is not None else "") + "by a primal-dual interior point, "
                        + ("inner-loop backtracking with later trials evaluated for the problems still searching only"
                           if cfg["nx"] * (cfg["nx"] + cfg["nu"]) >= 12 else
                           "deferred backtracking (one callback launch per iteration -- blocks, defects and tiles of the trial point; an "
                           "accepted trial's evaluation is the next iterate's), parallel-in-time LQ solve")
                        + ", unconverged problems compacted to the front as the batch converges; mpc_solved_per_s counts status == 0 only (this rank's iteration "
                        "statistics)"}

    def gather_latency_us(self, res, reps=200):
        """isolated latency of one u0 all-gather (stream-ordered, HIP events)"""
        eng, cfg = res["eng"], res["cfg"]
        if self.dist is None:
            return None
        torch = self.torch
        B = res["B"]
        Zd = torch.zeros(B, eng.n, dtype=eng.dtype, device=self.dev)
        if eng.comm is not None:
            fn = lambda: eng.allgather_u0(Z=Zd)                            # noqa: E731
        else:
            from pyneuralempc_amd.parallel import allgather_u0, first_controls
            fn = lambda: allgather_u0(first_controls(Zd, cfg["H"], cfg["nx"], cfg["nu"]), total=self.world * B)  # noqa: E731
        self.barrier()
        t = self.timed_events(fn, reps, prime_ms=0)
        return self.max_over_ranks(t) * 1e6

    def emit(self, full):
        """rank 0: detailed record to the side file and stderr, then the driver's line as the LAST stdout line"""
        if self.rank != 0:
            return
        write_details(full, name=self.args.details_file)
        sys.stdout.flush()
        os.dup2(self.saved_stdout, 1)
        print(driver_line(full), flush=True)
        os.dup2(2, 1)

    # -------------------------------------------------------------------------------------------------------------
    def main(self):
        np, torch = self.np, self.torch
        args = self.args
        if args.only_hessian:
            # profiling mode: the Hessian-callback legs of one configuration alone, so that a rocprofv3 kernel trace of
            # this command averages those launches only
            res = self.run_config(args.config, 5, 2, headline=False, kernel=args.kernel)
            out = {"config": {"workload": res["cfg"]["label"], "batch_per_gpu": res["B"]},
                   "hessian_callback": self.hessian_leg(res)}
            self.emit(out)
            return
        if args.only_sparse:
            # profiling mode: the sparse-contract leg of one configuration alone (rocprofv3 kernel trace of ONE launch shape)
            res = self.run_config(args.config, 5, 2, headline=False, kernel=args.kernel)
            out = {"config": {"workload": res["cfg"]["label"], "batch_per_gpu": res["B"]}, "sparse_contract": self.sparse_leg(res)}
            self.emit(out)
            return
        res = self.run_config(args.config, args.steps, args.warmup, headline=True, kernel=args.kernel)
        cfg, B, eng, wall = res["cfg"], res["B"], res["eng"], res["wall"]
        primary, secondary, rf_extra = self.roofline_of(res)
        out = {
            "metric": "nlp_callback_evals_per_sec (f + grad f + g + dense jac g)",
            "value": self.world * B * args.steps / wall,
            "unit": "problem-evals/s",
            "n_gpus": self.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["label"], "batch_per_gpu": B, "H": cfg["H"], "nx": cfg["nx"], "nu": cfg["nu"],
                       "hidden": cfg["hidden"], "integrator": cfg["integrator"], "n": eng.n, "m": eng.m,
                       "row_kernel": eng.kernel_variant, "parallelism": f"problem-sharded x{self.world}",
                       # `value` is a SUSTAINED-CLOCK rate: the timed region is run once as the process finds the GPU
                       # (value_from_cold_gpu) and again after prime_ms of the same step, untimed (value)
                       "clock": "sustained (timed region after prime_ms of the same step)" if args.prime_ms > 0 else "cold",
                       "prime_ms": res.get("primed_ms", 0.0),
                       "value_from_cold_gpu": (self.world * B * args.steps / res["wall_cold"]) if res.get("wall_cold") else None,
                       "latency_bound_note": ("B=256 is 320 tiles on 256 CUs: launch latency, not throughput, sets the time"
                                              if B * cfg["H"] <= 16 * 2 * 256 else None)},
            "batch_evals_per_s": args.steps / wall,
            "jacobian_max_abs_err_vs_cpu": res["errs"]["jac"], "max_abs_err_vs_cpu": res["errs"],
            "err_checked_on": f"outputs of the timed B={B} launch, problems {res['checked_slices']} (+16 each), rank {self.rank}",
            "roofline": primary,
            "roofline_" + secondary["bound"] + ("_whole_eval" if secondary["bound"] == "hbm" else "_row_kernel"): secondary,
            **rf_extra,
            "eval_us": {"timed_loop": wall / args.steps * 1e6, "event_loop": res["t_all"] * 1e6,
                        "p10_median_p90": res["step_pcts"]},
            "value_from_cold_gpu": (self.world * B * args.steps / res["wall_cold"]) if res.get("wall_cold") else None,
            "clock_priming": {"ms": res.get("primed_ms", 0.0),
                              "note": "untimed run of the same step before the W warm-up steps: an idle GPU starts below its "
                                      "sustained clock (--prime-ms 0 measures from cold)"},
        }
        if self.dist is None and not args.only_eval:
            out["allgather_u0"] = self.single_rank_gather_latency(res)
        if self.dist is not None:
            out["allgather_u0"] = {
                "per_mpc_step_every_n_evals": max(1, args.evals_per_mpc_step), "issued_in_timed_loop": res["n_gather"],
                "path": "nempc_allgather_u0 (libnempc.so -> RCCL ncclAllGather), on the launch stream" if eng.comm is not None
                        else "torch.distributed/" + self.backend,
                "latency_us": self.gather_latency_us(res), "rows_gathered": self.world * B}
            if getattr(self, "comm_error", None):
                out["allgather_u0"]["cabi_error"] = self.comm_error
        # HBM bytes of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py):
        # FETCH_SIZE * 2 (gfx950 correction) + WRITE_SIZE per launch, tools/summarize_profiles.py.  The headline roofline
        # takes the entry of the FUSED instantiation (template argument FUSE = true), the kernel the timed loop launches.
        pmc_file = os.path.join(REPO, "profiles", args.pmc_file)
        if args.config == "c2" and B == 1024 and eng.kernel_variant == "mfma" and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            for k, v in pmc.items():
                if not k.startswith(str(res["row_kernel"])) or "hbm_traffic_bytes" not in v:
                    continue
                is_fused = ", true" in k
                targets = ([out["roofline"], secondary] if is_fused else [rf_extra.get("roofline_row_kernel_unfused", {})])
                for rf in targets:
                    if not rf:
                        continue
                    rf["traffic"] = v["hbm_traffic_bytes"]
                    rf["traffic_note"] = ("FETCH_SIZE*2 + WRITE_SIZE per launch of " + k + ", profiles/" + args.pmc_file +
                                          (" (the fused evaluation: algorithmic 0.5 MB read, 20.5 MB written)" if is_fused else
                                           " (algorithmic: 0.51 MB read, 1.3 MB written)"))

        if not args.only_eval:
            # ---- two independent batches in flight (reported apart; `value` is the single-stream figure)
            if self.rank == 0:
                out["pipelined_two_streams"] = self.two_stream_leg(res)
            # ---- batched solver + all-gather of the solved u0 (collective: every rank)
            if cfg["integrator"] != "rk4" or eng.kernel_variant != "valu":
                out["batched_solver"] = self.solver_leg(res)
            if not args.no_hessian and (cfg["integrator"] != "rk4" or eng.kernel_variant != "valu"):
                out["hessian_callback"] = self.hessian_leg(res)
            out["sparse_contract"] = self.sparse_leg(res)
            # ---- the other BASELINE configs under the same clock (HIP events; every rank runs them, rank 0 reports)
            if args.config == "c2" and not args.batch and not args.no_other_configs:
                others = {}
                names = ["c2_b256", "c3", "c5"] + (["c4"] if self.world > 1 else [])
                for nm in names:
                    r2 = self.run_config(nm, 50, 5, headline=False)
                    p2, s2, x2 = self.roofline_of(r2)
                    c2 = r2["cfg"]
                    entry = {"workload": c2["label"], "batch_per_gpu": r2["B"], "dtype": c2["dtype"],
                             "ms_per_step": r2["t_all"] * 1e3,
                             "batch_evals_per_s_per_gpu": 1.0 / r2["t_all"],
                             "problem_evals_per_s": self.world * r2["B"] / self.max_over_ranks(r2["t_all"]),
                             "roofline": p2, "max_abs_err_vs_cpu": r2["errs"], "max_abs_ref": r2["err_scale"],
                             "checked_slices": r2["checked_slices"]}
                    if r2["t_all_rotating"]:
                        entry["ms_per_step_rotating_outputs"] = r2["t_all_rotating"] * 1e3
                        entry["rotation"] = r2["rotation"]
                    entry.update(x2)
                    if s2["bound"] == "mfma":
                        entry["roofline_mfma"] = s2
                    if not args.no_hessian and nm != "c4":
                        entry["hessian_callback"] = self.hessian_leg(r2)
                    if nm != "c4":
                        entry["sparse_contract"] = self.sparse_leg(r2)
                    if nm == "c4":
                        entry["batched_solver"] = self.solver_leg(r2)
                    others[nm] = entry
                    del r2
                out["other_configs"] = others
            if self.world == 1 and args.config == "c2" and not args.batch and not args.no_other_configs:
                out["layered_path"] = self.layered_leg()
                out["steady_state"] = self.steady_state_leg(cfg)
                out["narrow_networks"] = self.narrow_leg()
                out["shard_c4"] = self.shard_c4_leg()
                out["solver_c3"] = self.solver_c3_leg()
            if self.world == 1 and not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(cfg)
        self.emit(out)
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()

    def solver_c3_leg(self):
        """The batched solver at configs[2]'s dims (6/3, MLP 3 x 128, H = 30, RK4, fp32, B = 1024; bounds |x| <= 3, |u| <= 0.5; the
        set-up of tools/solver_profile.py c3): share of the batch converged within 40 / 80 outer iterations and the time of the
        solve (review item: >= 85 % at 40)."""
        import time
        np, torch = self.np, self.torch
        from oracle import nempc_oracle as orc
        from pyneuralempc_amd import CallbackEngine
        nx, nu, H, B = 6, 3, 30, 1024
        net = orc.MLP.random(nx + nu, [128, 128, 128], nx, seed=0)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="rk4", DT=0.1, dtype=torch.float32, device=self.dev, max_batch=B)
        X0 = eng.to_device(np.random.default_rng(100).uniform(-0.5, 0.5, size=(B, nx)))
        lb = np.concatenate([np.full(H * nx, -3.0), np.full(H * nu, -0.5)])
        out = {"workload": "B=1024, H=30, 6/3, MLP 3x128 tanh, RK4, f32; |x| <= 3, |u| <= 0.5"}
        for mi in (40, 80):
            eng.solve(X0, lb=lb, ub=-lb, max_iter=3)
            torch.cuda.synchronize(self.dev)
            t = time.perf_counter()
            Z, st, it = eng.solve(X0, lb=lb, ub=-lb, max_iter=mi)
            torch.cuda.synchronize(self.dev)
            dt = time.perf_counter() - t
            ok = int((st == 0).sum())
            out[f"budget_{mi}"] = {"iterations": int(it), "converged_frac": ok / B, "solve_ms": dt * 1e3, "mpc_solved_per_s": ok / dt}
        return out

    def layered_leg(self):
        """Networks outside the register-resident kernels (the reference wraps any feed-forward Keras model:
        model/tensorflow.py:8-29): the layer-at-a-time GEMM path (csrc/kernels_layered.hip) on the headline's dims with a
        2 x 256 and a 3 x 256 tanh network in fp64, and on configs[2]'s dims (6/3, H = 30, RK4, fp32) with a 4 x 512 one -- whole evaluation (f, grad, g, dense Jacobian) and exact-Hessian callback,
        HIP-event time, fraction of the FP64 matrix peak by algorithmic GEMM flops, error of the timed launches' own output
        against the oracle on the first problems."""
        np, torch = self.np, self.torch
        from oracle import nempc_oracle as orc
        from pyneuralempc_amd import CallbackEngine
        out = {}
        for name, hidden, (B, H, nx, nu), integ, DT, tdt in (("2x256", [256, 256], (1024, 20, 2, 1), "discret", 1.0, torch.float64),
                                                            ("3x256", [256, 256, 256], (1024, 20, 2, 1), "discret", 1.0, torch.float64),
                                                            # configs[2]'s dims and precision with a 4 x 512 network
                                                            ("4x512_rk4", [512] * 4, (1024, 30, 6, 3), "rk4", 0.1, torch.float32)):
            S = 4 if integ == "rk4" else 1
            f64 = tdt == torch.float64
            net = orc.MLP.random(nx + nu, hidden, nx, seed=0)
            prob = orc.Problem(net, H, nx, nu, orc.RK4 if S == 4 else orc.DISCRET, DT, Q=np.eye(nx), R=0.1 * np.eye(nu))
            eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=tdt, device=self.dev, max_batch=B)
            eng.set_objective(Q=np.eye(nx), R=0.1 * np.eye(nu))
            Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            rng = np.random.default_rng(7)
            lamh, sigh = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
            lam, sig = eng.to_device(lamh), eng.to_device(sigh)
            step, outs = eng.bind(Z, X0, ("f", "grad", "g", "jac_dense"))
            t_e = self.timed_events(step, 30 if S == 1 else 8)
            call_h, out_h = eng.bind_hess(Z, X0, lam, sig)
            t_h = self.timed_events(call_h, 20 if S == 1 else 4)
            torch.cuda.synchronize(self.dev)
            dims = [nx + nu] + hidden + [nx]
            F = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:]))
            row_flops = B * H * S * (1 + nx) * F
            hess_flops = B * H * S * (2 + nx + nu) * F + (row_flops if S == 4 else 0)
            peak = PEAK_F64_TFLOPS if f64 else PEAK_F32_TFLOPS
            e_j = e_h = s_h = 0.0
            for i in range(4 if S == 1 else 1):           # (the 6/3 RK4 oracle with 512-wide layers takes seconds per problem)
                f, grad, g, J = prob.eval_batch(Zh[i:i + 1], X0h[i:i + 1])
                e_j = max(e_j, float(np.abs(outs["jac_dense"][i].to("cpu", torch.float64).numpy() - J[0]).max()),
                          float(np.abs(outs["g"][i].to("cpu", torch.float64).numpy() - g[0]).max()))
                ref = prob.hessian_values(Zh[i], X0h[i], lamh[i], sigh[i])
                e_h = max(e_h, float(np.abs(out_h["hvals"][i].to("cpu", torch.float64).numpy() - ref).max()))
                s_h = max(s_h, float(np.abs(ref).max()))
            out[name] = {"workload": f"B={B}, H={H}, {nx}/{nu}, MLP {name.split('_')[0]} tanh, {'RK4' if S == 4 else 'Discret'}, {'f64' if f64 else 'f32'}",
                         "kernel_variant": eng.kernel_variant,
                         "evaluation": {"us": t_e * 1e6, "kernel": eng.last_row_kernel, "gflop": row_flops / 1e9,
                                        "frac_of_matrix_peak": row_flops / t_e / 1e12 / peak, "max_abs_err_vs_cpu": e_j},
                         "hessian_callback": {"us": t_h * 1e6, "kernel": eng.last_hess_kernel, "gflop": hess_flops / 1e9,
                                              "frac_of_matrix_peak": hess_flops / t_h / 1e12 / peak,
                                              "flops_note": "(2 + nin) GEMM sweeps per row" + (" and RK4 stage + the stage-record rows launch" if S == 4 else "")
                                                            + "; the per-layer contraction is vector work and not counted",
                                              "max_abs_err_vs_cpu": e_h, "max_abs_ref": s_h}}
            del eng
        return out

    def narrow_leg(self):
        """Networks of width <= 128 with a per-layer activation mix or a fourth hidden layer (the reference wraps any
        feed-forward Keras model, model/tensorflow.py:8-29): since round 5 on the register-resident matrix-core kernels with
        run-time activation codes (before: the layered path).  Headline dims (2/1, H = 20, B = 1024); rows launch (g + tiles)
        and exact-Hessian callback, HIP events, fraction of the matrix peak by (1 + nx) / (2 + nin) network passes."""
        np, torch = self.np, self.torch
        from oracle import nempc_oracle as orc
        from pyneuralempc_amd import CallbackEngine
        B, H, nx, nu = 1024, 20, 2, 1
        out = {}
        for name, hidden, acts, tdt in (("3x128_mix_f32", [128] * 3, ["relu", "tanh", "sigmoid", "linear"], torch.float32),
                                        ("4x64_tanh_f32", [64] * 4, "tanh", torch.float32),
                                        ("4x64_tanh_f64", [64] * 4, "tanh", torch.float64),
                                        ("2x64_mix_f64", [64] * 2, ["tanh", "elu", "linear"], torch.float64)):
            net = orc.MLP.random(nx + nu, hidden, nx, seed=0, activations=acts)
            eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=tdt, device=self.dev, max_batch=B, activations=net.act)
            Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
            Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
            lamh = np.random.default_rng(7).normal(size=(B, eng.m))
            step, outs = eng.bind(Z, X0, ("g", "jac_tiles"))
            t = self.timed_events(step, 50)
            ch, outh = eng.bind_hess(Z, X0, eng.to_device(lamh), eng.to_device(np.ones(B)))
            th = self.timed_events(ch, 20)
            torch.cuda.synchronize(self.dev)
            prob = orc.Problem(net, H, nx, nu)
            _, _, g, _ = prob.eval_batch(Zh[:4], X0h[:4])
            err = float(np.abs(outs["g"][:4].to("cpu", torch.float64).numpy() - g).max())
            ref = prob.hessian_values(Zh[0], X0h[0], lamh[0], 1.0)
            errh = float(np.abs(outh["hvals"][0].to("cpu", torch.float64).numpy() - ref).max())
            dims = [nx + nu] + hidden + [nx]
            F = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:]))
            peak = PEAK_F64_TFLOPS if tdt == torch.float64 else PEAK_F32_TFLOPS
            out[name] = {"kernel_variant": eng.kernel_variant,
                         "evaluation": {"us": t * 1e6, "kernel": eng.last_row_kernel, "max_abs_err_vs_cpu": err,
                                        "frac_of_matrix_peak": B * H * (1 + nx) * F / t / 1e12 / peak},
                         "hessian_callback": {"us": th * 1e6, "kernel": eng.last_hess_kernel, "max_abs_err_vs_cpu": errh,
                                              "frac_of_matrix_peak": B * H * (2 + nx + nu) * F / th / 1e12 / peak}}
            del eng
        return out

    def steady_state_leg(self, cfg, B=8192):
        """The headline kernel away from the launch-latency share: the same one-launch evaluation at B = 8192 (8 x the
        tiles, the ~4 us of dispatch / first-round-trip / tail are then 1/8 of the share they are at B = 1024)."""
        from oracle import nempc_oracle as orc
        eng = self.make_engine(cfg, B)
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1)
        step, _ = eng.bind(eng.to_device(Zh), eng.to_device(X0h), ("f", "grad", "g", "jac_dense"))
        t = min(self.timed_events(step, 100) for _ in range(3))        # (best of three: 160 MB of output per launch, the first repetition runs 15 % slow on some boxes)
        work = algorithmic_work(cfg, B, eng.m, eng.n)
        peak = PEAK_F64_TFLOPS if cfg["dtype"] == "f64" else PEAK_F32_TFLOPS
        return {"batch": B, "us": t * 1e6, "frac": work["flops"] / t / 1e12 / peak,
                "hbm_frac_dense_contract": work["dense_bytes"] / t / 1e9 / PEAK_HBM_GBS, "kernel": str(eng.last_row_kernel)}

    def shard_c4_leg(self):
        """configs[3]'s per-rank shard on ONE GPU (B = 4096 / 8 = 512 problems of the configs[2] problem): the time a rank of
        the 8-GPU job spends per evaluation, on record while no 8-GPU node has run the job."""
        from oracle import nempc_oracle as orc
        cfg = dict(CONFIGS["c4"], batch=512)
        B = 512
        eng = self.make_engine(cfg, B)
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1)
        step, _ = eng.bind(eng.to_device(Zh), eng.to_device(X0h), ("f", "grad", "g", "jac_dense"))
        t = self.timed_events(step, 50)
        work = algorithmic_work(cfg, B, eng.m, eng.n)
        return {"batch_per_rank": B, "us": t * 1e6, "frac": work["flops"] / t / 1e12 / PEAK_F32_TFLOPS,
                "problem_evals_per_s": B / t, "kernel": str(eng.last_row_kernel),
                "note": "one rank's share of configs[3] (global batch 4096 over 8 GPUs), dense contract, HIP events"}

    def two_stream_leg(self, res):
        torch = self.torch
        from oracle import nempc_oracle as orc
        cfg, B, eng = res["cfg"], res["B"], res["eng"]
        want = ("f", "grad", "g", "jac_dense")
        eng2 = self.make_engine(cfg, B, self.args.kernel)
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1 + self.rank)
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        Zb, X0b = (eng2.to_device(a) for a in orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=101 + self.rank))
        sa, sb = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        with torch.cuda.stream(sa):
            step_a, _ = eng.bind(Z, X0, want)
        with torch.cuda.stream(sb):
            step_b, _ = eng2.bind(Zb, X0b, want)
        torch.cuda.synchronize(self.dev)
        for k in range(20):
            (step_a if k % 2 == 0 else step_b)()
        torch.cuda.synchronize(self.dev)
        steps = self.args.steps
        tp = time.perf_counter()
        for k in range(steps):
            (step_a if k % 2 == 0 else step_b)()
        torch.cuda.synchronize(self.dev)
        tp = (time.perf_counter() - tp) / steps
        del eng2
        return {"us_per_eval": tp * 1e6, "batch_evals_per_s": 1.0 / tp,
                "note": "two handles on two HIP streams, independent batches alternating (this rank only)"}

    def hessian_leg(self, res):
        """The Hessian callbacks of the configuration (reference: IpoptProblem.hessian, optimizer/ipopt.py:66-86; the
        Gauss-Newton variant is BASELINE.json's north_star / configs[4]): HIP-event time, roofline and the error of the
        timed launch's own output against the oracle on three 16-problem slices.
        Algorithmic work.  Exact blocks: base forward + base reverse sweep, then one tangent forward + one tangent reverse
        sweep per network input: (2 + 2 nin) network passes per row and RK4 stage against the row kernel's (1 + nx), plus
        -- RK4 -- the stage-record row launch itself.  Gauss-Newton: the row kernel's flops (tiles), then B * nnz values."""
        np, torch = self.np, self.torch
        cfg, B, eng = res["cfg"], res["B"], res["eng"]
        from oracle import nempc_oracle as orc
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1 + self.rank)
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        rng = np.random.default_rng(7 + self.rank)
        lamh, sigh = rng.normal(size=(B, eng.m)), rng.uniform(0.5, 1.5, size=B)
        wh = rng.uniform(0.2, 1.5, size=(B, cfg["H"] * cfg["nx"]))
        lam, sig, wgt = eng.to_device(lamh), eng.to_device(sigh), eng.to_device(wh)
        reps = max(self.args.steps // 4, 10)
        nx, nin, S = cfg["nx"], cfg["nx"] + cfg["nu"], (4 if cfg["integrator"] == "rk4" else 1)
        peak_tf = PEAK_F64_TFLOPS if cfg["dtype"] == "f64" else PEAK_F32_TFLOPS
        w = 8 if cfg["dtype"] == "f64" else 4
        row_flops = res["work"]["flops"]
        hess_flops = row_flops * (2 + 2 * nin) / (1 + nx) + (row_flops if S == 4 else 0)
        # (bound calls: at B = 256 a callback is 10 us of device time, less than the checked Python wrapper costs per call)
        call_h, out_h = eng.bind_hess(Z, X0, lam, sig)
        t_h = self.timed_events(call_h, reps)
        hv = out_h["hvals"].clone()
        call_g, out_g = eng.bind_hess(Z, X0, wgt, sig, gauss_newton=True)
        t_g = self.timed_events(call_g, reps)
        gv = out_g["hvals"].clone()
        torch.cuda.synchronize(self.dev)
        _, prob = oracle_problem(cfg)
        k = min(16, B)
        starts = sorted({0, max(0, B // 2 - k // 2), B - k})
        e_h = e_g = s_h = 0.0
        for s0 in starts:
            idx = range(s0, s0 + k) if cfg["nx"] <= 2 else range(s0, s0 + min(k, 4))     # (the 6/3 RK4 oracle is slow)
            for i in idx:
                ref = prob.hessian_values(Zh[i], X0h[i], lamh[i], sigh[i])
                e_h = max(e_h, float(np.abs(hv[i].to("cpu", torch.float64).numpy() - ref).max()))
                s_h = max(s_h, float(np.abs(ref).max()))
                refg = prob.gauss_newton_values(Zh[i], X0h[i], wh[i], sigh[i])
                e_g = max(e_g, float(np.abs(gv[i].to("cpu", torch.float64).numpy() - refg).max()))
        out_bytes = B * eng.nnz_hess * w
        # flops the launched kernel EXECUTES on the matrix cores: the compiled-shape kernel (rowhess_coopfx_kernel) sums the
        # second-order chain rule layer by layer -- base forward + base reverse + nin tangent-forward sweeps = (2 + nin)
        # network passes; the forward-over-reverse kernels (rowhess_coop / rowhess_mfma) run a tangent-forward AND a
        # tangent-reverse sweep per input = (2 + 2 nin) passes.  `frac` is taken over the executed count.
        hk = eng.last_hess_kernel or ""
        passes_fwd_over_rev = 2 + 2 * nin
        passes = (2 + nin) if hk.endswith("rowhess_coopfx_kernel") else passes_fwd_over_rev
        stage_rows = row_flops if S == 4 else 0
        flops_exec = row_flops * passes / (1 + nx) + stage_rows
        flops_f_o_r = row_flops * passes_fwd_over_rev / (1 + nx) + stage_rows
        kname = (hk + " (blocks and tril assembly in one launch)" if S == 1 else
                 "RK4 pipeline: stage-record rows, rk4_nu, " + hk.replace("rk4:", "") + " (direct mode), rk4_congruence, assemble_hess")
        gn_fused = eng.last_row_kernel == "rows_coopfx_kernel"
        return {"nnz_hess": eng.nnz_hess,
                "exact": {"us": t_h * 1e6, "batch_evals_per_s": 1.0 / t_h, "max_abs_err_vs_cpu": e_h, "max_abs_ref": s_h,
                          "roofline": {"bound": "mfma", "kernel": kname,
                                       "achieved": flops_exec / t_h / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                                       "frac": flops_exec / t_h / 1e12 / peak_tf, "traffic": None,
                                       "flops_per_callback": flops_exec,
                                       "flops_note": f"matrix-core flops the launched kernel executes: ({passes} network passes per row"
                                                     + (" and RK4 stage" if S == 4 else "") + ") / (1 + nx) x the row kernel's flops"
                                                     + (" + the stage-record row launch" if S == 4 else "")
                                                     + "; the contraction over the hidden units is vector work and not counted",
                                       "frac_vs_fwd_over_rev_equivalent": flops_f_o_r / t_h / 1e12 / peak_tf,
                                       "fwd_over_rev_note": "the same time priced at forward-over-reverse's (2 + 2 nin) passes: what "
                                                            "the callback would cost in flops without the layer-wise factorisation"}},
                "gauss_newton": {"us": t_g * 1e6, "batch_evals_per_s": 1.0 / t_g, "max_abs_err_vs_cpu": e_g,
                                 "roofline": {"bound": "mfma", "kernel": ("rows_coopfx_kernel<..., GN = true>: tiles, blocks and tril assembly in one launch"
                                                                          if gn_fused else
                                                                          str(eng.last_row_kernel) + " (tiles) + assemble_hess_gn_kernel: two launches"),
                                              "achieved": row_flops / t_g / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                                              "frac": row_flops / t_g / 1e12 / peak_tf, "traffic": None,
                                              "flops_per_callback": row_flops, "bytes_written": out_bytes,
                                              "hbm_frac_of_output": out_bytes / t_g / 1e9 / PEAK_HBM_GBS}},
                "err_checked_on": f"hvals of the timed launches, problems {starts} (+{k if cfg['nx'] <= 2 else 4} each)"}

    def sparse_leg(self, res):
        """The sparse contract (what a real Ipopt run wants; the reference hands cyipopt the dense (m, n),
        optimizer/ipopt.py:88-96): f, grad, g and the band-pattern Jacobian values in nempc_jac_structure order from ONE
        launch -- no dense matrix, no tile round trip, no assembly launch.  HIP-event time, both rooflines over the
        compact bytes of SURVEY 8d (inputs, f, grad, g, band values), error of the timed launch's output vs the oracle."""
        np, torch = self.np, self.torch
        cfg, B, eng = res["cfg"], res["B"], res["eng"]
        from oracle import nempc_oracle as orc
        Zh, X0h = orc.synthetic_inputs(B, cfg["H"], cfg["nx"], cfg["nu"], seed=1 + self.rank)
        Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
        step, outs = eng.bind(Z, X0, ("f", "grad", "g", "jac_sparse"))
        t = self.timed_events(step, max(self.args.steps, 50))
        kern = eng.last_row_kernel
        rows, cols = eng.jac_structure()
        _, prob = oracle_problem(cfg)
        k = min(16, B)
        starts = sorted({0, max(0, B // 2 - k // 2), B - k})
        err = {"f": 0.0, "grad": 0.0, "g": 0.0, "jac_sparse": 0.0}
        for s0 in starts:
            sl = slice(s0, s0 + (k if cfg["nx"] <= 2 else min(k, 4)))
            f, grad, g, jac = prob.eval_batch(Zh[sl], X0h[sl])
            for key, ref in (("f", f), ("grad", grad), ("g", g), ("jac_sparse", jac[:, rows, cols])):
                err[key] = max(err[key], float(np.abs(outs[key][sl].to("cpu", torch.float64).numpy() - ref).max()))
        w = 8 if cfg["dtype"] == "f64" else 4
        peak_tf = PEAK_F64_TFLOPS if cfg["dtype"] == "f64" else PEAK_F32_TFLOPS
        nnz = int(len(rows))
        nbytes = B * w * ((eng.n + cfg["nx"]) + 1 + eng.n + eng.m + nnz)
        flops = res["work"]["flops"]
        ai, ridge = flops / nbytes, peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        mfma = {"bound": "mfma", "kernel": str(kern), "achieved": flops / t / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / peak_tf, "traffic": None, "kernel_us": t * 1e6, "flops_per_launch": flops,
                "arithmetic_intensity": ai, "ridge": ridge}
        hbm = {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "traffic": None, "bytes_per_launch": nbytes}
        return {"us": t * 1e6, "batch_evals_per_s": 1.0 / t, "problem_evals_per_s": B / t, "nnz_jac": nnz,
                "one_launch": kern in ("rows_coopfx_kernel+sparse", "rows_coop_kernel+sparse"),
                "dense_contract_us": res["t_all"] * 1e6,
                "roofline": mfma if ai >= ridge else hbm, ("roofline_hbm" if ai >= ridge else "roofline_mfma"): hbm if ai >= ridge else mfma,
                "max_abs_err_vs_cpu": err, "err_checked_on": f"outputs of the timed launch, problems {starts}"}

    def single_rank_gather_latency(self, res, reps=200):
        """N = 1: no gather is issued in the timed loop (there is nobody to exchange with), but the path exists -- a
        one-rank RCCL communicator on the handle -- and its isolated latency is reported so that the N > 1 lines, which
        issue it once per MPC step inside the loop, can be read against it."""
        eng = res["eng"]
        info = {"issued_in_timed_loop": 0, "n_gather": 0, "per_mpc_step_every_n_evals": max(1, self.args.evals_per_mpc_step),
                "rows_gathered": res["B"], "path": "nempc_allgather_u0 (libnempc.so -> RCCL ncclAllGather), one-rank communicator",
                "latency_us": None}
        try:
            from pyneuralempc_amd import CallbackEngine
            if eng.comm is None:
                eng.comm_init(1, 0, CallbackEngine.comm_unique_id())
            Zd = self.torch.zeros(res["B"], eng.n, dtype=eng.dtype, device=self.dev)
            info["latency_us"] = self.timed_events(lambda: eng.allgather_u0(Z=Zd), reps, prime_ms=0) * 1e6
        except Exception as e:      # noqa: BLE001  (RCCL absent: the evaluation metric does not depend on it)
            info["error"] = f"{type(e).__name__}: {e}"
        return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (default: the config's)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "valu", "mfma", "mfma_tile"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2-B256 / C3 / C5 legs of the default run")
    ap.add_argument("--only-eval", action="store_true",
                    help="profiling mode: the timed loop, the row-kernel timing and the accuracy check of the timed "
                         "launch only (no two-stream, solver, Hessian, other-config or CPU legs), so that a rocprofv3 "
                         "kernel trace of this command averages the headline launch alone")
    ap.add_argument("--only-hessian", action="store_true",
                    help="profiling mode: the Hessian-callback legs (exact + Gauss-Newton) of --config only")
    ap.add_argument("--only-sparse", action="store_true",
                    help="profiling mode: the sparse-contract leg (f, grad, g, band values in one launch) of --config only")
    ap.add_argument("--hessian", action="store_true", help="(kept for old command lines: the Hessian legs run by default)")
    ap.add_argument("--no-hessian", action="store_true", help="skip the Hessian-callback legs (exact Lagrangian + Gauss-Newton)")
    ap.add_argument("--evals-per-mpc-step", type=int, default=17,
                    help="N > 1: one u0 all-gather per this many callback evaluations inside the timed loop")
    ap.add_argument("--prime-ms", type=float, default=40.0,
                    help="untimed run of the headline step before the warm-up steps, to reach the sustained clock (0: off)")
    ap.add_argument("--solver-iters", type=int, default=64)
    ap.add_argument("--pmc-file", default="r05_c2_b1024_pmc.json")
    ap.add_argument("--details-file", default=None, help="name of the detailed record (default: bench_details.json)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher role: nothing above this line has touched HIP (torch is not even imported yet)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries exactly one JSON line: libraries that print banners while they initialise (RCCL does, to stdout) are
    # sent to stderr by pointing fd 1 there until the line is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = Rank(args)
    rank.saved_stdout = saved_stdout
    rank.main()


if __name__ == "__main__":
    main()
