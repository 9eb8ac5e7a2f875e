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
