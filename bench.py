#!/usr/bin/env python3
"""Headline benchmark: PivotCVAE train step (fwd + bwd + Adam [+ gradient all-reduce]) on synthetic data.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched as one rank per GPU by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment), or started plainly - then this process spawns `python -m torch.distributed.run --nproc-per-node N`
on itself BEFORE it touches the GPU, passes the ranks' output through and exits with their code.

Metric (BASELINE.json): slates/sec (+ the ELBO terms) at catalog N=1M, slate K=10, emb D=128, global batch
B=8192, variant pivotcvae_gt_pi, full-catalog softmax (n_neg = N), Z=16, hidden 256/256, prior 128/128.
One "step" = one pass of the hot path over one global batch; inputs are resident in HBM before the timed
region.  With N GPUs the global batch is sharded (B/N slates per rank, "strong" scaling), replicas are kept
in sync by ONE RCCL all-reduce of the flat gradient buffer per step.

OUTPUT CONTRACT (round 6).  The LAST line of stdout is ONE JSON object of at most LINE_LIMIT (6144) bytes - `headline_line()`:
metric / value / ms_per_step / config / elbo, and
  roofline     - the dominant kernel (fused catalog softmax-CE) priced against the dense MFMA peak of the
                 arithmetic it runs in; its duration is measured live with HIP events on the launch stream,
                 inside the timed steps;
  cpu_baseline - the CPU oracle (a port of the reference's torch-CPU train step, dense [B*S, N] logits)
                 timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only);
  parity       - HIP-vs-oracle relative error of the ELBO terms on the baseline sample;
  dist         - (N > 1 or a forced 1-rank group) the all-reduce's own time and the dominant kernel's min / max over ranks;
  summaries    - a few two-number summaries of the side blocks.
Everything else - the verbose roofline notes, `variants`, `pivot_rules`, `mlp_roofline`, `gather_roofline`, `generate`,
`validation`, `pretrain_env`, `epoch`, `arithmetic_error_vs_fp64` (bench_extras.py) - goes to a side FILE, `bench_extras.json` next to
this script (or --extras-file / $PCVAE_BENCH_EXTRAS), never to stdout.  tests/test_bench_launch.py holds the line to that contract
on a stub result, tests/test_hip_bench_rehearsal.py on the real run.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LINE_LIMIT = 6144   # bytes of the final stdout line (the driver keeps an 8 KB tail of stdout: round 5's 20 KB line was cut)

CONFIGS = {
    # name: (N items, S slots, D emb, B global batch)   SURVEY.md section 8(d)
    "1": dict(N=1_000, S=5, D=16, B=64, model="listcvae"),   # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "2": dict(N=10_000, S=5, D=32, B=1024),
    "3": dict(N=100_000, S=10, D=64, B=4096),
    "4": dict(N=1_000_000, S=10, D=128, B=8192),
    "5": dict(N=10_000_000, S=20, D=256, B=8192),   # + response-model in-loop evaluation ("eval" block)
}
Z, H, HP, N_USER = 16, 256, 128, 10_000
BETA, LR = 0.001, 3e-4
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}


def structs(S, D, model="pivotcvae_gt_pi"):
    C = S + 1
    if model == "listcvae":
        return dict(enc=[S * D + C + D, H, H], dec=[Z + C + D, H, H, S * D], prior=[C + D, HP, HP])
    return dict(enc=[S * D + C + D, H, H], psm=[Z + C + D, H, H, D], scm=[Z + C + 2 * D, H, H, (S - 1) * D],
                prior=[C + D, HP, HP])


def build_model(cfg, device, dtype):
    import pivotcvae_amd as pa
    N, S, D = cfg["N"], cfg["S"], cfg["D"]
    torch.manual_seed(0)  # weight seed 0 (reference init scheme: kaiming_uniform_ weights, default biases)
    a = (2.0 / D) ** 0.5
    gen = torch.Generator(device=device).manual_seed(0)
    doc = torch.nn.Embedding(N, D, device=device)
    doc.weight.data = (torch.rand(N, D, device=device, generator=gen) * 2 - 1) * a  # env/response_model.py:29-31
    usr = torch.nn.Embedding(N_USER, D, device=device)
    usr.weight.data = (torch.rand(N_USER, D, device=device, generator=gen) * 2 - 1) * a
    st = structs(S, D, "listcvae" if cfg.get("model") == "listcvae" else "pivotcvae")
    if cfg.get("model") == "listcvae":
        from pivotcvae_amd.models.listcvae import UserListCVAEWithPrior
        m = UserListCVAEWithPrior(doc, usr, S, D, Z, S + 1, st["enc"], st["dec"], st["prior"], False, device)
    else:
        m = pa.PIVOTCVAE_MODELS[cfg.get("model", "pivotcvae_gt_pi")](doc, usr, S, D, Z, S + 1, st["enc"], st["psm"], st["scm"],
                                                                    st["prior"], False, device)
    m.set_catalog_precision(dtype)
    return m, st


def synthetic_batch(cfg, B, device, seed=1):
    g = torch.Generator(device=device).manual_seed(seed)  # data seed 1
    s = torch.randint(0, cfg["N"], (B, cfg["S"]), device=device, generator=g)
    u = torch.randint(0, N_USER, (B, 1), device=device, generator=g)
    r = (torch.rand(B, cfg["S"], device=device, generator=g) < 0.5).float()
    return s, r, u


def kernel_name(R, N, D, dtype):
    """the dominant kernel's name as rocprofv3 shows it (the library reports which variant a shape runs)"""
    from pivotcvae_amd import _hip
    from pivotcvae_amd import ops
    if dtype in ("bf16x3", "bf16x6") and ops.split_width(_hip.PREC_NAMES[dtype], D):
        D = ops.split_width(_hip.PREC_NAMES[dtype], D)    # narrower tables run the 128-wide kernel on zero columns
    v = _hip.lib().pcvae_catalog_ce_variant(R, N, D, _hip.PREC_NAMES[dtype])
    return {0: f"catalog_ce_f32_kernel<{D}>", 1: f"catalog_ce_bf16_fast_kernel<{D}>",
            2: f"catalog_ce_bf16_pipe_kernel<{D}, {2 if D == 256 else 4}>", 3: f"catalog_ce_x3_pipe_kernel<{D}, {1 if D == 256 else 2}, 2>",
            4: f"catalog_ce_x3_pipe_kernel<{D}, 2, 3>"}.get(v, "?")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_candidates(model, st, cfg, n_candidate):
    """The reference's DEFAULT mode on the host cores: the oracle's candidate branch (embedding [Bs, S, Cn, D] + bmm + CE + backward +
    Adam) on a bounded sample, the candidate sets drawn the reference's way (numpy randint per slate + the first-hit / overwrite rule,
    data_loader.py:46-58), + HIP-vs-oracle ELBO on the same sets."""
    import numpy as np
    from oracle import pivotcvae_oracle as orc
    S, D, N = cfg["S"], cfg["D"], cfg["N"]
    Bs = max(8, min(cfg["B"], int(2.5e8 // (S * n_candidate * D))))      # the gathered rows [Bs, S, Cn, D] fp32 stay under 1 GB
    steps = 5
    dev = model.docEmbed.weight.device
    s, r, u = synthetic_batch(cfg, Bs, dev, seed=11)
    eps = torch.randn(Bs, Z, generator=torch.Generator().manual_seed(2))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ocfg = orc.Config(cfg.get("model", "pivotcvae_gt_pi"), S, D, Z, False, st)
    sc, rc, uc = s.cpu(), r.cpu(), u.cpu()
    np.random.seed(3)
    t0 = time.perf_counter()
    raw = np.stack([np.random.randint(N, size=(S, n_candidate)) for _ in range(Bs)])
    cand, tgt = orc.candidate_targets(sc, torch.from_numpy(raw))
    t_draw = time.perf_counter() - t0
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(32, ncpu))
    kw = dict(candidates=cand, cand_targets=tgt)
    (ol, orec, okld), grads = orc.loss_and_grads(sd, ocfg, sc, rc, uc, eps, BETA, **kw)
    state, cur = {}, sd
    t0 = time.perf_counter()
    for _ in range(steps):
        _, g = orc.loss_and_grads(cur, ocfg, sc, rc, uc, eps, BETA, **kw)
        cur = orc.adam_step(cur, g, state, LR)
    dt = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        hl, hrec, hkld = model.loss(s, r, u, BETA, eps=eps.to(dev), candidates=(cand.to(dev), tgt.to(dev)))
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    base = {"value": Bs / (dt + t_draw), "unit": "slates/s", "cores": torch.get_num_threads(), "host_cores": ncpu, "cpu_model": cpu_model(),
            "kind": "port", "as_specified": Bs == cfg["B"],
            "sample": f"oracle train step in candidate mode (rows [{Bs}, {S}, {n_candidate}, {D}] + bmm + CE + KL + backward + Adam: "
                      f"{dt:.3f} s/step) + the reference's host-side candidate draw ({t_draw:.3f} s), B={Bs} slates, {steps} steps"}
    parity = {"loss_rel_err": rel(hl.item(), ol), "recLoss_rel_err": rel(hrec.item(), orec), "KLD_rel_err": rel(hkld.item(), okld),
              "tolerance": 1e-4, "sample": f"B={Bs}, same eps, same candidate sets, HIP fused candidate kernel vs CPU oracle"}
    parity["within_tolerance"] = max(parity["loss_rel_err"], parity["recLoss_rel_err"], parity["KLD_rel_err"]) <= 1e-4
    return base, parity


def cpu_baseline_and_parity(model, st, cfg, dtype):
    """Oracle train step on the host cores on a bounded sample + HIP-vs-oracle ELBO on that same sample."""
    from oracle import pivotcvae_oracle as orc
    # [Bs*S, N] fp32 logits + its autograd temporaries must fit host RAM: config 4: 160 x 1M x 4 B = 640 MB each.  Configs 1 and 2
    # run AS SPECIFIED (B = 64 / 1024: 1.3 MB / 205 MB of logits - the sizes the reference itself runs on a CPU).
    as_is = cfg["B"] * cfg["S"] * cfg["N"] <= 64e6
    Bs = cfg["B"] if as_is else max(1, min(16, int(160e6 // (cfg["S"] * cfg["N"]))))
    steps = 5
    dev = model.docEmbed.weight.device
    s, r, u = synthetic_batch(cfg, Bs, dev, seed=11)
    eps = torch.randn(Bs, Z, generator=torch.Generator().manual_seed(2))  # eps seed 2
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ocfg = orc.Config(cfg.get("model", "pivotcvae_gt_pi"), cfg["S"], cfg["D"], Z, False, st)
    sc, rc, uc = s.cpu(), r.cpu(), u.cpu()
    # torch's CPU ops do not scale to every hardware thread of a big host (256 threads: 30x SLOWER than 32 on
    # the dual EPYC 9575F box, tools/cpu_threads_probe.py); give the baseline its best thread count
    ncpu = os.cpu_count() or 1
    best_t, best_dt = 1, float("inf")
    for th in sorted({min(t, ncpu) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        q = max(1, Bs // 4)
        orc.loss_and_grads(sd, ocfg, sc[:q], rc[:q], uc[:q], eps[:q], BETA)
        t0 = time.perf_counter()
        orc.loss_and_grads(sd, ocfg, sc[:q], rc[:q], uc[:q], eps[:q], BETA)
        if time.perf_counter() - t0 < best_dt:
            best_t, best_dt = th, time.perf_counter() - t0
    torch.set_num_threads(best_t)
    state = {}
    (ol, orec, okld), grads = orc.loss_and_grads(sd, ocfg, sc, rc, uc, eps, BETA)  # warm-up at the full sample
    t0 = time.perf_counter()
    cur = sd
    for _ in range(steps):
        _, g = orc.loss_and_grads(cur, ocfg, sc, rc, uc, eps, BETA)
        cur = orc.adam_step(cur, g, state, LR)
    dt = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        hl, hrec, hkld = model.loss(s, r, u, BETA, eps=eps.to(dev))
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    base = {"value": Bs / dt, "unit": "slates/s", "cores": torch.get_num_threads(), "host_cores": ncpu,
            "cpu_model": cpu_model(), "kind": "port", "as_specified": as_is,
            "sample": f"oracle/pivotcvae_oracle.py train step (dense [{Bs * cfg['S']},{cfg['N']}] logits + CE + KL + "
                      f"backward + Adam), " + ("the config AS SPECIFIED: " if as_is else "") +
                      f"B={Bs} slates of the same workload, {steps} steps, {dt:.3f} s/step"}
    parity = {"loss_rel_err": rel(hl.item(), ol), "recLoss_rel_err": rel(hrec.item(), orec),
              "KLD_rel_err": rel(hkld.item(), okld), "tolerance": 1e-4,
              "sample": f"B={Bs}, same eps, HIP {dtype} vs CPU oracle"}
    parity["within_tolerance"] = max(parity["loss_rel_err"], parity["recLoss_rel_err"], parity["KLD_rel_err"]) <= 1e-4
    return base, parity


def self_launch(n, argv):
    """--gpus N > 1 without a launcher: start one rank per GPU through torch.distributed.run and pass their output through.
    Runs BEFORE anything in this process touches the GPU (importing torch and counting devices do not); the ranks are
    child processes, this process only waits for them and exits with their code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = subprocess.call(cmd, env=env)
    sys.stdout.flush()
    raise SystemExit(rc)


def dry_run(args, world, rank):
    """PCVAE_BENCH_DRYRUN=1: the launch / rendezvous / reduce-and-print skeleton on gloo without any GPU work (CPU test of the
    N > 1 path: tests/test_bench_launch.py)."""
    import torch.distributed as dist
    B = args.global_batch or CONFIGS[args.config]["B"]
    if B % world:
        raise SystemExit("global batch not divisible by the number of GPUs")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or os.environ.get("PCVAE_BENCH_FORCE_DIST") == "1":
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ranks = dist.get_world_size()
        dist.destroy_process_group()
    else:
        t, ranks = torch.tensor([1.0]), 1
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks": t.item(), "config": {"global_batch": B, "per_gpu_batch": B // world}}), flush=True)


class StepTimer:
    """the contract's timed region: W untimed steps, then exactly K steps between barrier + synchronize on both sides, MAX over
    ranks; HIP events on the launch stream around the dominant kernel inside those steps, and - where a process group exists -
    around the gradient all-reduce (eager, outside the replayed graph: `Trainer.REDUCE_TIMING`)"""

    def __init__(self, trainer, batch, B, lo, use_dist, device):
        self.tr, self.batch, self.B, self.lo, self.use_dist, self.device = trainer, batch, B, lo, use_dist, device

    def sync_all(self):
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup):
        import torch.distributed as dist
        from pivotcvae_amd import ops
        tr = self.tr
        s, r, u = self.batch
        events = []

        def hook_begin():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            return e0, e1

        def hook_end(pair):
            pair[1].record()
            events.append(pair)

        pivot_events, reduce_events = [], []

        def pivot_end(pair):
            pair[1].record()
            pivot_events.append(pair)

        def reduce_end(pair):
            pair[1].record()
            reduce_events.append(pair)

        trace = [] if os.environ.get("PCVAE_BENCH_TRACE_ELBO") == "1" else None   # debugging aid: every step's terms on stderr
        for _ in range(warmup):
            if trace is not None:
                trace.append(tr.step(s, r, u, global_batch=self.B, row_offset=self.lo))
            else:
                tr.step(s, r, u, global_batch=self.B, row_offset=self.lo)
        # capture_graph set = the steps WILL be replayed (captured in the first step if --warmup 0 left that to the timed region): no
        # event hooks then - an event record inside a capture is an error (hipErrorInvalidHandle)
        tr.prepare_graph(s, r, u, self.lo)   # --warmup 0: the one-off capture happens HERE, never inside the timed region
        graphed = bool(tr.capture_graph)
        # every rank must launch its steps the same way: a world where some ranks replay a graph and others fell back to eager
        # launches would only show up as a slow, meaningless timing - fail loudly, on every rank (all see the same two numbers)
        extra = graphed
        if self.use_dist:
            flag = torch.tensor([1.0 if graphed else 0.0, -1.0 if graphed else 0.0], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            any_graphed, all_graphed = bool(flag[0].item() > 0), bool(flag[1].item() < 0)
            if any_graphed != all_graphed:
                raise SystemExit(f"rank {dist.get_rank()}: hipGraph capture succeeded on some ranks and failed on others "
                                 f"(this rank: {'captured' if graphed else 'eager: ' + str(tr.capture_failed)}); refusing to time a mixed world")
            extra = any_graphed
        if not graphed:
            ops.CATALOG_CE_TIMING = (hook_begin, hook_end)
            ops.PIVOT_TIMING = (hook_begin, pivot_end)
        if self.use_dist:   # the all-reduce is always an eager launch between the (replayed) local phase and Adam
            tr.REDUCE_TIMING = (hook_begin, reduce_end)
        self.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, rec, kld = tr.step(s, r, u, global_batch=self.B, row_offset=self.lo)
            if trace is not None:
                trace.append((loss, rec, kld))
        self.sync_all()
        dt = time.perf_counter() - t0
        tr.REDUCE_TIMING = None
        if trace is not None:
            for i, t in enumerate(trace):
                print(f"[elbo trace] step {i}: " + " ".join(f"{float(v):.6f}" for v in t), file=sys.stderr, flush=True)
        ops.CATALOG_CE_TIMING = None
        ops.PIVOT_TIMING = None
        dt_local = dt
        if self.use_dist:
            t = torch.tensor([dt], device=self.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        if extra:
            # HIP events cannot be recorded inside a hipGraph (ROCm 7.2: hipErrorInvalidHandle, tools/evt_graph_probe.py),
            # so the dominant kernel is timed over the same number of EAGER steps right after the timed region:
            # same kernel, same inputs, same launch stream.
            was = tr.capture_graph
            tr.capture_graph = False
            ops.CATALOG_CE_TIMING = (hook_begin, hook_end)
            ops.PIVOT_TIMING = (hook_begin, pivot_end)
            for _ in range(steps):
                tr.step(s, r, u, global_batch=self.B, row_offset=self.lo)
            torch.cuda.synchronize()
            ops.CATALOG_CE_TIMING = None
            ops.PIVOT_TIMING = None
            tr.capture_graph = was
        kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
        # the pivot-selection kernel of a pt / spt / sgt train step (catalog argmax / rejection sampler), per step
        pivot_ms = sum(a.elapsed_time(b) for a, b in pivot_events) / max(steps, 1) if pivot_events else None
        graphed = graphed and tr._graph is not None and tr.capture_failed is None   # what actually happened, for the line's label
        out = dict(dt=dt, steps=steps, kern_ms=kern_ms, pivot_ms=pivot_ms, graphed=graphed, elbo=(loss, rec, kld))
        if self.use_dist:
            # the collective's own time (event pair on the launch stream around all_reduce: the stream waits for RCCL's stream on both
            # sides, so the pair spans hand-over + transfer), and how evenly the ranks were loaded: min / max over ranks of the
            # dominant kernel and of the local step time
            red_ms = sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1)
            hi = torch.tensor([red_ms, kern_ms, dt_local * 1e3 / steps, -red_ms, -kern_ms, -dt_local * 1e3 / steps],
                              device=self.device, dtype=torch.float64)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            hi = hi.tolist()
            out["dist"] = {"allreduce_ms": hi[0], "allreduce_ms_min": -hi[3], "allreduce_share_of_step": hi[0] / (dt * 1e3 / steps),
                           "kernel_ms_max": hi[1], "kernel_ms_min": -hi[4], "rank_step_ms_max": hi[2], "rank_step_ms_min": -hi[5],
                           "allreduce_timed_over": f"{len(reduce_events)} all-reduces inside the timed steps, one HIP event pair each "
                                                   "on the launch stream; MAX (and min) over ranks of the per-rank mean"}
        return out


# what the catalog contraction computes in, per --dtype: (json dtype, MFMA peak it is priced against, MFMAs issued per
# algorithmic MAC, one-phrase description for the line)
ARITH = {"f32": ("f32", PEAK_TFLOPS["f32"], 1, "f32 (v_mfma_f32_32x32x2_f32)"),
         "bf16": ("bf16", PEAK_TFLOPS["bf16"], 1, "bf16 operands, fp32 accumulate"),
         "bf16x3": ("bf16x3", PEAK_TFLOPS["bf16"], 3, "bf16x3: operands as bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate (2^-18 per product; a "
                                                      "stated-tolerance fast path, narrower than fp32)"),
         "bf16x6": ("bf16x6", PEAK_TFLOPS["bf16"], 6, "bf16x6: every fp32 operand as 3 bf16 components (sum = the fp32 value), 6 bf16 MFMAs per "
                                                      "product, fp32 accumulate - the reference's fp32 arithmetic on the bf16 matrix cores")}


# uniformly random row gathers, chip-wide (MI355X_MICROARCH.md "Indexed rows: gather into LDS"): rows served by the XCD's own L2,
# by the Infinity Cache / fabric (tables of 38 .. 302 MB: 7.4 - 8.6 TB/s, "the table's size costs 10 %, all of it L2 share"; past
# 256 MiB 3 - 9 % over that curve), and the HBM peak for the bytes that MUST come from memory
GATHER_L2_TBPS, GATHER_FABRIC_TBPS, HBM_TBPS = 17.8, 8.6, 8.0


def sparse_roofline(name, R_local, N, D, kern_ms, sparse_kept, traffic=None, elem_bytes=4):
    """The sparse (n_neg << N) kernel is a row gather: R (n_neg + 1) rows of 4 D bytes REQUESTED, out of a table of only 4 N D bytes
    - every table row is re-read ~R n_neg / N times, so most requests are served by the caches, not by HBM, and pricing the
    requested bytes against the HBM peak is not a roofline (round 2 did: 0.93 at config 4, 1.10 at config 3).  The bound here is a
    TIME: the compulsory bytes (the table once, if fewer bytes than requested; rx in, nll / lse / dx out) at the HBM peak + the
    re-read bytes at the guide's measured uniformly-random gather rates (an XCD's 4 MiB L2 holds 4 MiB / T of a table of T bytes;
    the rest comes over the fabric from the Infinity Cache / HBM).  frac = that time / the measured time, always <= 1 unless the
    kernel beats the guide's gather loop.  `traffic` = the L2's memory-side bytes of one launch from the committed rocprofv3
    FETCH_SIZE / WRITE_SIZE passes (profiles/traffic.json; on gfx950 Infinity-Cache hits are included in it)."""
    requested = float(R_local) * sparse_kept * D * elem_bytes   # (elem_bytes = 2: the bf16-row variants of configs 3 / 5)
    table = float(N) * D * elem_bytes
    compulsory = min(table, requested) + float(R_local) * (2 * D * 4 + 8 + 8)   # table once + rx read + dx written + nll, lse
    reread = max(requested - min(table, requested), 0.0)
    h = min(1.0, 4.0 * 2 ** 20 / table)   # share of uniformly random requests an XCD's L2 serves
    t_bound = compulsory / (HBM_TBPS * 1e12) + reread * ((1 - h) / (GATHER_FABRIC_TBPS * 1e12) + h / (GATHER_L2_TBPS * 1e12))
    t = kern_ms * 1e-3
    ach = requested / t / 1e9 if t > 0 else 0.0
    eff_peak = requested / t_bound / 1e9
    return {"kernel": name, "bound": "cache+hbm gather model", "achieved": ach, "peak": eff_peak, "unit": "GB/s",
            "frac": ach / eff_peak if eff_peak else 0.0, "hbm_frac": ach / (HBM_TBPS * 1e3),
            "hbm_frac_note": "requested bytes / s against the 8 TB/s HBM spec: NOT a roofline for this kernel (every table row is "
                             "re-read from the caches many times; it may exceed 1), kept for readers that expect an HBM figure",
            "traffic": traffic, "ms_per_launch": kern_ms, "algorithmic_bytes_per_launch": requested,
            "compulsory_hbm_bytes_per_launch": compulsory, "hbm_frac_of_compulsory": compulsory / t / (HBM_TBPS * 1e12) if t > 0 else 0.0,
            "peak_model": f"requested bytes / (compulsory bytes at {HBM_TBPS} TB/s HBM + re-read bytes at the guide's random-row gather "
                          f"rates: {GATHER_FABRIC_TBPS} TB/s fabric / Infinity Cache, {GATHER_L2_TBPS} TB/s for the L2 share "
                          f"{h:.3f} of a {table / 1e6:.0f} MB table); the HBM peak alone is NOT this kernel's roof: "
                          f"{requested / 1e9:.1f} GB are requested out of a {table / 1e9:.2f} GB table"}


def candidate_roofline(R_local, N, D, Cn, kern_ms, traffic=None, bf16_rows=False):
    """the fused candidate-set kernel is the same uniformly random row gather as the sparse kernel - R (Cn + 1) rows of 4 D bytes
    requested (Cn candidates + the target row once more for the gradient) - priced on the same cache + HBM gather-time model"""
    out = sparse_roofline(f"candidate_ce_kernel<{D}, true, {'true' if bf16_rows else 'false'}>", R_local, N, D, kern_ms, Cn + 1, traffic,
                          elem_bytes=2 if bf16_rows else 4)
    out["rows"] = "bf16 table rows widened exactly, fp32 products and sums" if bf16_rows else "fp32 table rows (the reference's arithmetic)"
    out["replaces"] = ("candidate_draw -> [R, Cn] int64 ids -> candidate_scores_kernel -> [R, Cn] p -> dense_ce_kernel -> [R, Cn] dp -> "
                       "candidate_scores_bwd_kernel (the reference: randint on the host + embedding [R, Cn, D] + bmm + CrossEntropyLoss "
                       "+ autograd); none of those arrays exists here")
    return out


def roofline_block(name, R_local, N, D, dtype, kern_ms, sparse_kept=None, traffic=None, bf16_rows=False):
    if sparse_kept is not None:
        return sparse_roofline(name, R_local, N, D, kern_ms, sparse_kept, traffic, elem_bytes=2 if bf16_rows else 4)
    flops = 4.0 * R_local * N * D   # logits 2RND + gradient direction 2RND
    _, peak, mult, _ = ARITH[dtype]
    ach = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    out = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
           "ms_per_launch": kern_ms, "algorithmic_flops_per_launch": flops}
    if mult != 1:
        out["mfma_issue_frac"] = ach * mult / peak
        out["note"] = (f"{mult} bf16 MFMAs per algorithmic multiply-add (one per kept pair of operand components): `frac` prices the "
                       f"ALGORITHMIC flops against the bf16 peak, `mfma_issue_frac` the MFMAs actually issued; in fp32 terms the "
                       f"algorithmic rate is {ach / PEAK_TFLOPS['f32']:.2f}x the dense f32-MFMA peak of {PEAK_TFLOPS['f32']} TFLOP/s")
        out["algorithmic_vs_f32_mfma_peak"] = ach / PEAK_TFLOPS["f32"]
        # the roof of THIS algorithm on the pipe it runs on: the dense bf16 peak divided by the MFMAs one multiply-add costs
        out["algorithm_peak_TFLOPs"] = peak / mult
        out["frac_of_algorithm_peak"] = ach * mult / peak
    return out


def pivot_block(cfg, B_local, pivot_ms, step_ms, rule="pt"):
    """the pivot-selection kernels of a step.  pt: the catalog argmax (bf16 screening + exact rescoring), 2 B N D algorithmic flops
    per step against the dense peak of the pipe it runs on.  spt / sgt: the reference scores all B x N pairs and hands the matrix to
    torch.multinomial; the rejection sampler draws from the same distribution with ~2 dot products per slate, so the 2 B N D flops
    are not done at all - no flop rate is quoted for it, only its time."""
    N, D = cfg["N"], cfg["D"]
    out = {"ms_per_step": pivot_ms, "share_of_step": pivot_ms / step_ms if step_ms else None}
    if rule == "pt":
        flops = 2.0 * B_local * N * D
        tf = flops / (pivot_ms * 1e-3) / 1e12 if pivot_ms else 0.0
        out.update({"kernel": "catalog_screen_pipe_kernel (bf16 screening) + exact fp32 rescoring", "algorithmic_TFLOPs": tf,
                    "frac_of_bf16_peak": tf / PEAK_TFLOPS["bf16"], "algorithmic_flops_per_step": flops})
    else:
        out.update({"kernel": "catalog_sample_reject_kernel (+ the Gumbel-max kernel's launch, whose workgroups leave at once: no row "
                              "was flagged)",
                    "replaces": f"the [B, N] score matrix + sigmoid + torch.multinomial of models/pivotcvae.py:349-351 "
                                f"({2.0 * B_local * N * D / 1e12:.2f} TFLOP per step): rejection sampling draws from exactly that "
                                "categorical with ~2 gathered rows per slate"})
    return out


def committed_traffic(key):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (PMC counters cannot be read from inside the run);
    None for a workload the passes were not collected on (--global_batch: another number of row blocks per launch)"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if key is None or not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get(key)


# ---------------------------------------------------------------------------------------------------------------------------
# the ONE line
# ---------------------------------------------------------------------------------------------------------------------------
def _short(v, digits=6):
    """floats to `digits` significant digits, recursively (the line is a summary: bench_extras.json keeps every digit)"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _short(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")
CONFIG_KEYS = ("workload", "model", "global_batch", "per_gpu_batch", "parallelism", "rccl_ranks", "rehearsal", "catalog_arithmetic",
               "mlp_arithmetic", "launch")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_launch", "mfma_issue_frac", "traffic", "timed_over")
CPU_KEYS = ("value", "unit", "cores", "host_cores", "kind", "as_specified", "sample")
PARITY_KEYS = ("loss_rel_err", "recLoss_rel_err", "KLD_rel_err", "tolerance", "within_tolerance", "sample")
DIST_KEYS = ("allreduce_ms", "allreduce_share_of_step", "kernel_ms_min", "kernel_ms_max", "rank_step_ms_min", "rank_step_ms_max")
# dropped in this order, one at a time, should a line ever exceed LINE_LIMIT (it does not: the stub test fills every field)
OPTIONAL_ORDER = ("extras_errors", "summaries", "pivot_kernel", "parity.sample", "cpu_baseline.sample", "roofline.timed_over",
                  "config.mlp_arithmetic", "config.catalog_arithmetic")


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def summaries_of(full):
    """two-number summaries of the side blocks for the line (each: the fraction of its roof + the time or rate it came from)"""
    out = {}
    g = full.get("gather_roofline") or {}
    if "frac" in g:
        out["gather"] = {"kernel": g.get("kernel"), "frac": g["frac"], "us_per_launch": g.get("us_per_launch")}
        t = g.get("train_step_kernel") or {}
        if "frac" in t:
            out["train_step_gather"] = {"kernel": t.get("kernel"), "frac": t["frac"], "us_per_launch": t.get("us_per_launch")}
    for key in ("mlp_roofline", "mlp_roofline_f32", "mlp_roofline_bf16x3", "mlp_roofline_bf16x6"):
        m = full.get(key) or {}
        if "frac" in m:
            out["mlp" if key == "mlp_roofline" else "mlp_" + key[len("mlp_roofline_"):]] = {"arithmetic": m.get("arithmetic"), "frac": m["frac"], "ms_per_step": m.get("ms_per_step")}
    gen = full.get("generate") or {}
    if "value" in gen:
        out["generate"] = {"slates_per_s": gen["value"], "frac": gen.get("frac"), "ids_identical_to_f32_kernel": gen.get("ids_identical_to_f32_kernel")}
    for name, v in (full.get("variants") or {}).items():
        if isinstance(v, dict) and "value" in v:
            out.setdefault("variants_slates_per_s", {})[name] = v["value"]
    ep = full.get("epoch") or {}
    for name, v in ep.items():
        if isinstance(v, dict) and "slates_per_s" in v:
            out.setdefault("epoch", {})[name] = {"slates_per_s": v["slates_per_s"], "loop_overhead_frac": v.get("loop_overhead_frac")}
    return out


def headline_line(full, extras_file=None):
    """`full` (everything this run measured) -> the ONE JSON line the driver parses, <= LINE_LIMIT bytes: the contract's keys, the
    config in short phrases, elbo, roofline, cpu_baseline, parity, dist, summaries.  Pure function of `full` (tested on a stub)."""
    line = _pick(full, LINE_KEYS)
    cfg = _pick(full.get("config", {}), CONFIG_KEYS)
    for k, n in (("workload", 260), ("catalog_arithmetic", 220), ("mlp_arithmetic", 120), ("launch", 120), ("rehearsal", 120)):
        if k in cfg:
            cfg[k] = _clip(cfg[k], n)
    line["config"] = cfg
    if "elbo" in full:
        line["elbo"] = full["elbo"]
    if "roofline" in full:
        roof = _pick(full["roofline"], ROOFLINE_KEYS)
        roof["kernel"] = _clip(roof.get("kernel", "?"), 100)
        if "timed_over" in roof:
            roof["timed_over"] = _clip(roof["timed_over"], 140)
        roof.setdefault("traffic", None)
        line["roofline"] = roof
    if "cpu_baseline" in full:
        line["cpu_baseline"] = _pick(full["cpu_baseline"], CPU_KEYS)
        line["cpu_baseline"]["sample"] = _clip(line["cpu_baseline"].get("sample", ""), 260)
    if "parity" in full:
        line["parity"] = _pick(full["parity"], PARITY_KEYS)
        line["parity"]["sample"] = _clip(line["parity"].get("sample", ""), 120)
    if "dist" in full:
        line["dist"] = _pick(full["dist"], DIST_KEYS)
    if "pivot_kernel" in full:
        line["pivot_kernel"] = _pick(full["pivot_kernel"], ("ms_per_step", "share_of_step", "frac_of_bf16_peak"))
    sm = summaries_of(full)
    if sm:
        line["summaries"] = sm
    errs = [k for k, v in full.items() if isinstance(v, dict) and set(v) == {"error"}]
    if errs:
        line["extras_errors"] = errs
    if extras_file:
        line["extras_file"] = os.path.basename(extras_file)
    line = _short(line)
    for key in OPTIONAL_ORDER:   # never needed so far; a guarantee, not a habit
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        a, _, b = key.partition(".")
        if b:
            line.get(a, {}).pop(b, None)
        else:
            line.pop(a, None)
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(text)} bytes > {LINE_LIMIT}")
    return text


def write_extras(full, path):
    """everything measured, verbose notes included -> the side file (never stdout)"""
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        return path
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr, flush=True)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="4", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16x6", "bf16x3", "bf16"],
                    help="arithmetic of the catalog contraction of the HEADLINE line.  Default: the reference's own arithmetic (fp32 "
                         "operands, fp32 accumulate) - config 4 (D = 128): bf16x6, every fp32 operand as three bf16 components = "
                         "exactly the fp32 value, six bf16 MFMAs per product; configs 1, 2: f32 (v_mfma_f32_32x32x2_f32; they are "
                         "launch-bound); configs 3 and 5 are stated in bf16 (BASELINE.json).  The other arithmetics - f32 always, "
                         "bf16x3 (a stated-tolerance fast path with 16-bit-mantissa operands, NOT fp32), bf16 - are measured as "
                         "named blocks under `variants` of the extras file")
    ap.add_argument("--mlp", default=None, choices=["f32", "bf16x3", "bf16x6"],
                    help="arithmetic of the MLP GEMMs of the train step.  Default: bf16x3 where the catalog contraction runs in bf16x3 "
                         "or bf16 (the whole step then computes on the bf16 matrix cores); bf16x6 (fp32-exact products on the bf16 "
                         "matrix cores) with --dtype bf16x6 and in the gather modes; exact f32 MFMA with --dtype f32")
    ap.add_argument("--n_neg", type=int, default=None, help="default: N (full-catalog softmax)")
    ap.add_argument("--n_candidate", type=int, default=None,
                    help="time the reference's DEFAULT training mode instead (no --mask_train: candidate sets of this many ids per "
                         "slot, data_loader.py:46-58, train_generative.py:52-57) - the fused candidate kernel")
    ap.add_argument("--model", default=None,
                    help="registry key of the model (any of PIVOTCVAE_MODELS: pivotcvae_{gt,pt,spt,sgt}_{pi,spi}); default: the "
                         "config's (pivotcvae_gt_pi; config 1: listcvae)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the mlp_roofline / gather_roofline / generate / eval / epoch blocks")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` blocks (other arithmetics, n_neg = 1000)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", action="store_true", help="replay a hipGraph at any batch size (default: only when the "
                                                         "per-rank batch is <= 4096 slates, where launches matter)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="override the config's global batch (e.g. 1024 on one GPU = the per-rank load of the 8-GPU run)")
    ap.add_argument("--extras-file", default=os.environ.get("PCVAE_BENCH_EXTRAS", os.path.join(ROOT, "bench_extras.json")),
                    help="where everything beside the headline line is written (JSON; default: bench_extras.json next to bench.py)")
    args = ap.parse_args()

    dry = os.environ.get("PCVAE_BENCH_DRYRUN") == "1"
    # PCVAE_BENCH_REHEARSAL=1: the N > 1 path with every rank on THE SAME GPU and gloo carrying the collectives (RCCL refuses two
    # ranks on one device) - a rehearsal of sharding, per-rank capture, step counts and the JSON line where only one GPU exists
    # (the builder's box); its throughput means nothing and the line says so
    rehearsal = os.environ.get("PCVAE_BENCH_REHEARSAL") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not dry and not rehearsal and torch.cuda.device_count() < args.gpus:   # counting devices does not initialise the GPU
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} visible")
        self_launch(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if dry:
        return dry_run(args, world, rank)

    import torch.distributed as dist
    from pivotcvae_amd import ops
    from pivotcvae_amd.train_generative import Trainer

    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("PCVAE_BENCH_FORCE_DIST") == "1"  # 1-rank RCCL group: exercises the N>1 code on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # PCVAE_BENCH_FORCE_DIST without a launcher: a 1-rank RCCL group
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            raise SystemExit(f"--gpus {args.gpus} / RANK {rank} but the process group has {dist.get_world_size()} ranks and calls "
                             f"this one {dist.get_rank()}")

    cfg = dict(CONFIGS[args.config])
    if args.global_batch:
        cfg["B"] = args.global_batch
    if args.model:
        import pivotcvae_amd as pa
        if args.model != "listcvae" and args.model not in pa.PIVOTCVAE_MODELS:
            raise SystemExit(f"--model {args.model}: one of listcvae, " + ", ".join(sorted(pa.PIVOTCVAE_MODELS)))
        cfg["model"] = args.model
    if args.n_candidate is not None and args.n_neg is not None:
        raise SystemExit("--n_candidate (candidate sets) and --n_neg (mask-train) are the two branches of get_gen_loss: pick one")
    N, S, D, B = cfg["N"], cfg["S"], cfg["D"], cfg["B"]
    if B % world:
        raise SystemExit("global batch not divisible by the number of GPUs")
    if args.dtype is None:
        # the reference's own arithmetic: fp32 operands, fp32 accumulation - on v_mfma_f32_32x32x2_f32 ("f32"), and where the
        # bf16x6 kernel exists and the catalog is large enough to be MFMA-bound (D = 128: config 4) on the bf16 matrix cores with
        # every fp32 operand carried exactly as three bf16 components; `variants.f32` is in the extras file
        args.dtype = {"3": "bf16", "5": "bf16"}.get(args.config, "bf16x6" if D == 128 else "f32")
    if args.dtype == "bf16x3" and ops.x3_width(D) is None:
        raise SystemExit(f"bf16x3 exists for D <= {ops.X3_MAX_PADDED}")
    if args.dtype == "bf16x6" and ops.x6_width(D) is None:
        raise SystemExit("bf16x6 exists for D <= 128")
    if args.mlp is None:
        args.mlp = ops.default_mlp_precision(args.dtype)
    model, st = build_model(cfg, device, args.dtype)
    model.set_mlp_precision(args.mlp)
    if args.dtype == "bf16":
        # configs 3 / 5 are STATED in bf16 (BASELINE.json): their gather kernels (candidate sets, n_neg << N, validation) read rows of
        # the bf16 table too.  An explicit switch - a bf16 catalog arithmetic alone leaves them on the fp32 table (round 6)
        model.set_gather_rows("bf16")
    # hipGraph replay pays off when the step is launch-bound (per-rank batch <= 4096 slates: ~50 launches of 5-30 us);
    # at a full single-GPU batch of config 4 the catalog kernel is > 95 % of the step and eager launches keep the HIP events
    # that time it inside the timed region
    # (the gather-bound modes - candidate sets, n_neg << N - are a few ms per step at any batch: launches matter there too)
    light = args.n_candidate is not None or (args.n_neg is not None and ops.sparse_ce_applies(args.n_neg / N, N))
    use_graph = (not args.no_graph) and (B // world <= 4096 or args.graph or light)
    # resident_batch: every step of the timed region passes the SAME unmodified tensors (inputs resident in HBM, as the contract
    # says), so a graph replay does not re-copy them into its static buffers
    trainer = Trainer(model, lr=LR, beta=BETA, n_neg=args.n_neg, capture_graph=use_graph, resident_batch=True,
                      n_candidate=args.n_candidate)
    s, r, u = synthetic_batch(cfg, B, device)
    (s, r, u), lo = trainer.shard(s, r, u)
    s, r, u = s.contiguous(), r.contiguous(), u.contiguous()
    timer = StepTimer(trainer, (s, r, u), B, lo, use_dist, device)
    R_local = s.shape[0] * S

    res = timer.run(args.steps, args.warmup)
    dt, kern_ms, graphed = res["dt"], res["kern_ms"], res["graphed"]
    loss, rec, kld = res["elbo"]
    sparse = args.n_neg is not None and ops.sparse_ce_applies(args.n_neg / N, N)
    cand_mode = args.n_candidate is not None
    # the gather kernels read fp32 rows unless bf16 rows were asked for (configs 3 / 5: the stated arithmetic is bf16)
    bf16_rows = ops.gather_rows_are_bf16(model)
    rows_dtype = "bf16 rows, fp32 accumulate" if bf16_rows else "f32"
    # the committed counter passes are of the configs as BASELINE states them: none for an overridden batch
    tkey = lambda what: None if args.global_batch else f"config{args.config}_{what}_gpus{world}"
    if cand_mode:
        roof = candidate_roofline(R_local, N, D, args.n_candidate, kern_ms,
                                  committed_traffic(tkey(f"cand{args.n_candidate}")), bf16_rows)
        sparse = True
    else:
        roof = roofline_block(kernel_name(R_local, N, D, args.dtype) if not sparse else "catalog_ce_sparse_kernel",
                              R_local, N, D, args.dtype, kern_ms, sparse_kept=(args.n_neg + 1) if sparse else None,
                              traffic=committed_traffic(tkey(f"nneg{args.n_neg}")) if sparse else None,
                              bf16_rows=bf16_rows)
    if not sparse:
        roof["kernel_note"] = "the events also span its row-bound prologue and merge kernels, <1% together"
        roof["traffic"] = committed_traffic(tkey(args.dtype))
    roof["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of this kernel, committed; not measured in this run)"
    roof["timed_over"] = (f"{args.steps} eager steps right after the timed graph-replayed steps (HIP events cannot be "
                          "recorded inside a hipGraph)") if graphed else "HIP events on the launch stream inside the timed steps"

    mname = cfg.get("model", "pivotcvae_gt_pi")
    out = {
        "metric": "slates/sec + ELBO, N=1M catalog K=10 B=8192" if args.config == "4" and not args.global_batch
                  else f"slates/sec config {args.config}",
        "value": B * args.steps / dt, "unit": "slates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": rows_dtype if (cand_mode or sparse) else ARITH[args.dtype][0], "data": "synthetic",
        "config": {"workload": f"{'ListCVAE' if mname == 'listcvae' else 'PivotCVAE ' + mname[10:]} train step (fwd+bwd+Adam), catalog N={N} slate K={S} emb D={D} "
                               f"global batch B={B}, " + (f"candidate sets of {args.n_candidate} ids per slot drawn in-kernel (the reference's default mode)"
                                                          if cand_mode else "full-catalog softmax" + ("" if args.n_neg is None else f" n_neg={args.n_neg}")),
                   "model": mname,
                   "global_batch": B, "per_gpu_batch": B // world, "parallelism": f"dp{world}",
                   "rccl_ranks": dist.get_world_size() if use_dist else 1,
                   **({"rehearsal": "all ranks on ONE GPU, gloo collectives: checks the N > 1 path, not its speed"} if rehearsal else {}),
                   "catalog_arithmetic": ("gather kernel: " + ("bf16 table rows widened exactly, " if bf16_rows else "fp32 table rows, ")
                                          + "fp32 fmaf dot products, fp32 online softmax")
                   if (cand_mode or sparse) else ARITH[args.dtype][3],
                   "mlp_arithmetic": ops.MLP_ARITHMETIC[args.mlp],
                   "launch": "hipGraph replay (zero-grad+fwd+bwd) + eager all-reduce + Adam" if graphed else "eager"},
        "elbo": {"loss": loss.item(), "recLoss": rec.item(), "KLD": kld.item()},
        "roofline": roof,
    }
    if "dist" in res:
        out["dist"] = res["dist"]
    if res.get("pivot_ms") is not None:
        out["pivot_kernel"] = pivot_block(cfg, B // world, res["pivot_ms"], dt / args.steps * 1e3, getattr(model, "TRAIN_RULE", "gt"))
    single = rank == 0 and world == 1
    if single and not args.no_cpu_baseline and mname in ("pivotcvae_gt_pi", "listcvae"):
        if cand_mode:
            out["cpu_baseline"], out["parity"] = cpu_baseline_candidates(model, st, cfg, args.n_candidate)
        else:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(model, st, cfg, args.dtype)
    if single and not (args.no_variants and args.no_extras):
        # bench_extras imports this module's helpers as `bench`: make that name this very module when run as a script
        sys.modules.setdefault("bench", sys.modules[__name__])
        import bench_extras
        ctx = dict(args=args, cfg=cfg, model=model, st=st, trainer=trainer, timer=timer, batch=(s, r, u), lo=lo, device=device,
                   use_dist=use_dist, R_local=R_local, bf16_rows=bf16_rows, rows_dtype=rows_dtype, headline_ms=out["ms_per_step"])
        out.update(bench_extras.run(ctx))
    if use_dist:
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio; push it out first so that the JSON line is the LAST line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stderr.flush()
    if rank == 0:
        path = write_extras(out, args.extras_file)
        print(headline_line(out, path), flush=True)


if __name__ == "__main__":
    main()
