"""bench.py --gpus N (N > 1) started WITHOUT a launcher must spawn its own ranks.

PCVAE_BENCH_DRYRUN=1 swaps the GPU work for the rendezvous / reduce / print skeleton on gloo, so the launch path -
argument pass-through, one process per rank, a single JSON line from rank 0, the exit code - runs on a CPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = dict(os.environ, PCVAE_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_spawns_its_own_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--global-batch", "512"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["max_over_ranks"] == 2.0   # the MAX over ranks really went through the collective
    assert out["config"] == {"global_batch": 512, "per_gpu_batch": 256}


def test_bench_single_process_needs_no_launcher():
    p = _run(["--gpus", "1", "--steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1


def test_a_failing_rank_fails_the_parent():
    # global batch not divisible by the world size -> the ranks raise SystemExit -> non-zero exit of the parent
    p = _run(["--gpus", "2", "--global-batch", "511"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


# ---- the ONE line (the 20 KB line of round 5 could not be parsed by the driver) ------------------------------------------
def _stub_full(n_variants=8, pad=400):
    """a `full` result as bench.main() assembles it, every block present and every free-text field LONGER than the line may carry"""
    long = "x" * pad
    roof = {"kernel": "catalog_ce_x3_pipe_kernel<128, 2, 3> " + long, "bound": "mfma", "achieved": 285.0773162149573, "peak": 2500.0,
            "unit": "TFLOP/s", "frac": 0.11403092648598293, "ms_per_launch": 147.1286476135254, "mfma_issue_frac": 0.6841855589158975,
            "traffic": 15529842790.4, "timed_over": long, "note": long, "peak_model": long, "traffic_source": long}
    side = {"kernel": "k " + long, "frac": 0.6366071234, "us_per_launch": 19.9200001, "note": long}
    return {
        "metric": "slates/sec + ELBO, N=1M catalog K=10 B=8192", "value": 55456.712345678, "unit": "slates/s", "n_gpus": 8, "steps": 20,
        "warmup": 5, "ms_per_step": 147.71912345, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16x6",
        "data": "synthetic",
        "config": {"workload": "PivotCVAE gt_pi train step " + long, "model": "pivotcvae_gt_pi", "global_batch": 8192,
                   "per_gpu_batch": 1024, "parallelism": "dp8", "rccl_ranks": 8, "rehearsal": long, "catalog_arithmetic": long,
                   "mlp_arithmetic": long, "launch": long},
        "elbo": {"loss": 13.914512345, "recLoss": 13.72301234, "KLD": 191.52112345},
        "roofline": roof,
        "dist": {"allreduce_ms": 0.0612345678, "allreduce_ms_min": 0.05, "allreduce_share_of_step": 0.0031234567, "kernel_ms_min": 18.1,
                 "kernel_ms_max": 18.9, "rank_step_ms_min": 19.1, "rank_step_ms_max": 19.7, "allreduce_timed_over": long},
        "pivot_kernel": {"ms_per_step": 0.027, "share_of_step": 0.0002, "kernel": long, "replaces": long},
        "cpu_baseline": {"value": 25.26081234, "unit": "slates/s", "cores": 32, "host_cores": 256, "cpu_model": long, "kind": "port",
                         "as_specified": False, "sample": long},
        "parity": {"loss_rel_err": 0.0, "recLoss_rel_err": 1.2345678e-8, "KLD_rel_err": 3.9264e-07, "tolerance": 1e-4,
                   "within_tolerance": True, "sample": long},
        "variants": {f"variant_{i}_{long[:30]}": {"value": 1e6 + i, "roofline": dict(roof), "elbo": {"loss": 1.0}} for i in range(n_variants)},
        "pivot_rules": {f"pivotcvae_{i}": {"train": {"value": 1.0, "note": long}} for i in range(4)},
        "arithmetic_error_vs_fp64": {"f32": {"lse_max_abs_err": 1e-6}, "note": long},
        "mlp_roofline": dict(side, arithmetic="bf16x6", ms_per_step=0.2),
        "mlp_roofline_f32": dict(side, arithmetic="f32", ms_per_step=0.52), "mlp_roofline_bf16x3": dict(side, arithmetic="bf16x3", ms_per_step=0.35),
        "gather_roofline": dict(side, train_step_kernel=dict(side), back_to_back=dict(side)),
        "generate": {"value": 448849.123, "frac": 0.505583, "ids_identical_to_f32_kernel": True, "f32_kernel": dict(side)},
        "validation": {"a": dict(side), "b": dict(side)}, "pretrain_env": dict(side),
        "epoch": {"candidates_1000": {"slates_per_s": 1.2e6, "loop_overhead_frac": 0.08, "note": long},
                  "mask_train_n_neg_1000": {"slates_per_s": 1.2e6, "loop_overhead_frac": 0.08}, "reference": long},
        "eval": {"error": "RuntimeError: " + long},
    }


def test_the_bench_line_is_one_bounded_parseable_json_object():
    sys.path.insert(0, ROOT)
    import bench
    full = _stub_full()
    assert len(json.dumps(full)) > 20000          # the round-5 situation: everything measured is far larger than the line may be
    line = bench.headline_line(full, "/somewhere/bench_extras.json")
    assert "\n" not in line and len(line) <= bench.LINE_LIMIT == 6144, len(line)
    out = json.loads(line)
    assert json.loads(json.dumps(out)) == out
    # the contract's keys, with the driver's types
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "elbo", "roofline", "cpu_baseline", "parity"):
        assert k in out, k
    assert out["value"] == 55456.7 and out["n_gpus"] == 8 and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert set(out["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "mfma_issue_frac"}
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-5
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert set(out["config"]) >= {"workload", "model", "global_batch", "per_gpu_batch", "parallelism", "rccl_ranks",
                                  "catalog_arithmetic", "mlp_arithmetic", "launch"}
    assert "model" in out["config"] and "variants" not in out and "pivot_rules" not in out and "validation" not in out
    assert set(out["dist"]) >= {"allreduce_ms", "kernel_ms_min", "kernel_ms_max"}
    assert out["summaries"]["gather"]["frac"] == 0.636607 and out["summaries"]["mlp"]["ms_per_step"] == 0.2
    assert out["summaries"]["epoch"]["candidates_1000"]["loop_overhead_frac"] == 0.08
    assert out["extras_errors"] == ["eval"] and out["extras_file"] == "bench_extras.json"
    # even an absurd number of side blocks cannot push the line over the limit: optional parts are dropped, the contract's stay
    big = bench.headline_line(_stub_full(n_variants=400, pad=3000))
    assert len(big) <= bench.LINE_LIMIT and {"roofline", "cpu_baseline", "value", "config"} <= set(json.loads(big))


def test_the_extras_go_to_a_file_not_to_stdout(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    full = _stub_full()
    path = bench.write_extras(full, str(tmp_path / "bench_extras.json"))
    assert json.load(open(path))["variants"].keys() == full["variants"].keys()
    assert bench.write_extras(full, str(tmp_path / "no_such_dir" / "x.json")) is None   # unwritable: reported on stderr, not fatal
