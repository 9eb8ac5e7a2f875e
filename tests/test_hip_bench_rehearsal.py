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
    assert two["config"]["launch"].startswith("hipGraph") and eager["config"]["launch"] == "eager"
    want = [one["elbo"][k] for k in ("loss", "recLoss", "KLD")]
    for d in (two, eager):   # five optimisation steps in: equal to rounding, replayed or eager, one rank or two
        np.testing.assert_allclose([d["elbo"][k] for k in ("loss", "recLoss", "KLD")], want, rtol=2e-5)
