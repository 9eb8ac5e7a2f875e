"""-m gpu: `bench.py --gpus 2` end to end with BOTH ranks on the one GPU of the box and gloo carrying the collectives
(PCVAE_BENCH_REHEARSAL=1; RCCL refuses two ranks per device): self-launch through torch.distributed.run, sharding, per-rank hipGraph
capture, equal step counts on every rank, rank 0's JSON line - and the ELBO terms of the 2-rank run equal the 1-rank run's at the same
step (train_generative.py:59-63: the mean over rows and the sum over slates do not depend on how the batch is split)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(gpus, extra):
    env = dict(os.environ, PCVAE_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", "2", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras", "--no-variants"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    assert out.stdout.rstrip("\n").splitlines()[-1] == lines[0] and len(lines[0]) <= 6144   # the LAST stdout line, bounded
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_report_the_single_rank_elbo():
    one = run_bench(1, [])
    two = run_bench(2, [])
    eager = run_bench(2, ["--no-graph"])
    for d, n in ((one, 1), (two, 2), (eager, 2)):
        assert d["n_gpus"] == n and d["config"]["global_batch"] == 1024 and d["config"]["per_gpu_batch"] == 1024 // n
        assert d["scaling"] == "strong" and d["steps"] == 3 and d["value"] > 0
    assert two["config"]["rccl_ranks"] == 2 and "rehearsal" in two["config"]
    # the collective's own time and the ranks' balance are in the N > 1 line, not in the 1-rank line without a group
    for d in (two, eager):
        assert set(d["dist"]) >= {"allreduce_ms", "allreduce_share_of_step", "kernel_ms_min", "kernel_ms_max", "rank_step_ms_min",
                                  "rank_step_ms_max"}
        assert d["dist"]["allreduce_ms"] > 0 and d["dist"]["kernel_ms_min"] <= d["dist"]["kernel_ms_max"]
        assert d["dist"]["rank_step_ms_max"] <= d["ms_per_step"] * 1.05
    assert "dist" not in one
    assert two["config"]["launch"].startswith("hipGraph") and eager["config"]["launch"] == "eager"
    want = [one["elbo"][k] for k in ("loss", "recLoss", "KLD")]
    for d in (two, eager):   # five optimisation steps in: equal to rounding, replayed or eager, one rank or two
        np.testing.assert_allclose([d["elbo"][k] for k in ("loss", "recLoss", "KLD")], want, rtol=2e-5)


@pytest.mark.timeout(1500)
def test_the_real_default_bench_prints_one_bounded_line_and_writes_the_extras(tmp_path):
    """round 5's line grew to 20 KB and the driver could not parse it.  The REAL default run (config 4 as BASELINE
    states it, every side block on, the CPU baseline included; 2 steps) must end stdout with ONE JSON object of <= 6144 bytes that
    carries the contract's keys + roofline + cpu_baseline + parity, and put everything else into the extras FILE."""
    extras = tmp_path / "bench_extras.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PCVAE_BENCH_EXTRAS=str(extras))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PCVAE_BENCH_REHEARSAL", "PCVAE_BENCH_DRYRUN", "PCVAE_BENCH_FORCE_DIST"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                         text=True, timeout=1400)
    assert out.returncode == 0, out.stderr[-3000:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) <= 6144, len(last)
    assert len(out.stdout) <= 8000, "nothing but the line (and a short RCCL banner at most) goes to stdout"
    d = json.loads(last)
    assert d["metric"] == "slates/sec + ELBO, N=1M catalog K=10 B=8192" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "bf16x6" and d["unit"] == "slates/s" and d["value"] > 0 and d["vs_baseline"] is None
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert d["config"]["global_batch"] == 8192 and d["config"]["model"] == "pivotcvae_gt_pi"
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["kernel"].startswith("catalog_ce_x3_pipe_kernel<128, 2, 3>") and roof["peak"] == 2500.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5 and 0.05 < roof["frac"] < 0.2 and roof["traffic"]
    assert 0.9 * d["ms_per_step"] < roof["ms_per_launch"] < d["ms_per_step"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["within_tolerance"] is True
    assert "extras_errors" not in d, d.get("extras_errors")
    sm = d["summaries"]
    assert sm["gather"]["frac"] > 0.4 and sm["mlp"]["frac"] > 0 and sm["generate"]["slates_per_s"] > 0
    full = json.load(open(extras))
    for k in ("variants", "pivot_rules", "mlp_roofline", "gather_roofline", "generate", "validation", "pretrain_env", "epoch",
              "arithmetic_error_vs_fp64"):
        assert k in full and "error" not in full[k], (k, full.get(k))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5)     # the file repeats the headline in full precision
    assert {"candidates_1000", "mask_train_n_neg_1000"} <= set(full["epoch"])


@pytest.mark.timeout(900)
def test_four_ranks_on_one_gpu_report_the_single_rank_elbo_in_candidate_mode():
    """the driver's scaling run goes to 4 and 8 ranks: the same rehearsal with FOUR ranks on the one GPU (gloo collectives), in the
    reference's default loss mode (candidate sets drawn in-kernel from streams keyed by GLOBAL slots): sharding by four, per-rank graph
    capture, the `dist` block, and ELBO terms equal to the single-rank run's"""
    one = run_bench(1, ["--n_candidate", "64"])
    four = run_bench(4, ["--n_candidate", "64"])
    assert four["n_gpus"] == 4 and four["config"]["per_gpu_batch"] == 256 and four["config"]["rccl_ranks"] == 4
    assert four["config"]["launch"].startswith("hipGraph") and four["roofline"]["kernel"].startswith("candidate_ce_kernel")
    assert four["dist"]["allreduce_ms"] > 0 and four["dist"]["kernel_ms_min"] <= four["dist"]["kernel_ms_max"]
    np.testing.assert_allclose([four["elbo"][k] for k in ("loss", "recLoss", "KLD")], [one["elbo"][k] for k in ("loss", "recLoss", "KLD")],
                               rtol=2e-5)
