"""The line bench.py hands the driver stays machine-readable: one strict-JSON line under 4 KB carrying the contract's
keys, `roofline` and `cpu_baseline`; everything else lives in the side file (round 4's 21 KB line left the driver with
`parsed: null`)."""
import copy
import json
import os

import pytest

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    """a detailed record shaped like bench.py's `out` (the round-4 driver-form record when the profile is there)"""
    p = os.path.join(REPO, "profiles", "r04_bench_driver_form.json")
    if os.path.exists(p):
        return json.load(open(p))
    rf = {"bound": "mfma", "kernel": "rows_coopfx_kernel<..., FUSE = true>: the launch of the timed loop (prose prose)",
          "achieved": 34.0, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.43, "traffic": 26.2e6, "kernel_us": 16.0}
    return {"metric": "nlp_callback_evals_per_sec (f + grad f + g + dense jac g)", "value": 6.4e7, "unit": "problem-evals/s",
            "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.016, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1] dims", "batch_per_gpu": 1024, "H": 20, "nx": 2, "nu": 1},
            "roofline": rf, "cpu_baseline": {"value": 1.8e5, "unit": "problem-evals/s", "cores": 16, "kind": "port",
                                              "sample": "x" * 400}}


def strict_loads(text):
    def refuse(tok):
        raise ValueError("non-standard JSON token " + tok)
    return json.loads(text, parse_constant=refuse)


def test_driver_line_is_small_strict_json_with_the_contract_keys():
    full = canned()
    text = bench.driver_line(full)
    assert "\n" not in text
    assert len(text.encode()) < 4096 == bench.DRIVER_LINE_MAX
    line = strict_loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["steps"] == full["steps"] and line["warmup"] == full["warmup"] and line["n_gpus"] == full["n_gpus"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    # no prose: every string of the line is short
    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(s) for s in strings(line)) <= 120


def test_driver_line_of_the_round_5_record_with_every_leg_fits():
    """the full round-5 record (layered, narrow, steady-state, shard and solver legs) still makes a line under 4 KB with its
    summary in place (not dropped)"""
    p = os.path.join(REPO, "profiles", "r05_bench_driver_form_details.json")
    if not os.path.exists(p):
        pytest.skip("no round-5 detailed record in profiles/")
    full = json.load(open(p))
    full.setdefault("solver_c3", {"workload": "x", "budget_40": {"iterations": 40, "converged_frac": 0.9727, "solve_ms": 71.5234567,
                                                                  "mpc_solved_per_s": 13926.123},
                                  "budget_80": {"iterations": 76, "converged_frac": 1.0, "solve_ms": 91.5234567, "mpc_solved_per_s": 11188.1}})
    text = bench.driver_line(full)
    assert len(text.encode()) < bench.DRIVER_LINE_MAX
    line = strict_loads(text)
    assert "dropped" not in line["summary"]
    for k in ("layered_2x256", "steady_state", "solver", "solver_c3", "c3", "c5", "shard_c4_b512"):
        assert k in line["summary"], k
    assert line["summary"]["solver_c3"]["40"]["converged_frac"] == pytest.approx(full["solver_c3"]["budget_40"]["converged_frac"], rel=1e-3)


def test_driver_line_never_carries_nan_or_infinity():
    full = copy.deepcopy(canned())
    full["roofline"]["traffic"] = float("nan")
    full["jacobian_max_abs_err_vs_cpu"] = float("inf")
    full["value_from_cold_gpu"] = float("-inf")
    text = bench.driver_line(full)
    line = strict_loads(text)
    assert line["roofline"]["traffic"] is None
    assert "NaN" not in text and "Infinity" not in text
    # the side file's form is strict, too
    strict_loads(json.dumps(bench._sanitize(full), allow_nan=False))


def test_an_oversized_summary_is_dropped_not_the_contract():
    full = copy.deepcopy(canned())
    rf = full["roofline"]
    full["other_configs"] = {
        f"cfg{i}": {"ms_per_step": 0.5, "roofline": rf, "max_abs_err_vs_cpu": {"jac": 1e-9},
                    "sparse_contract": {"us": 1.0, "roofline": rf},
                    "hessian_callback": {"exact": {"us": 1.0, "roofline": rf}, "gauss_newton": {"us": 1.0, "roofline": rf}}}
        for i in range(60)}
    text = bench.driver_line(full)
    assert len(text) < bench.DRIVER_LINE_MAX
    line = strict_loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert "dropped" in line["summary"]


def test_details_file_round_trip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    full = canned()
    bench.write_details(full, also_stderr=False)
    back = strict_loads(open(tmp_path / bench.DETAILS_FILE).read())
    assert back["metric"] == full["metric"] and back["roofline"]["frac"] == full["roofline"]["frac"]
