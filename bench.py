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

The JSON line also carries
  roofline     - the dominant kernel (fused catalog softmax-CE) priced against the dense MFMA peak of the
                 arithmetic it runs in; its duration is measured live with HIP events on the launch stream,
                 inside the timed steps;
  cpu_baseline - the CPU oracle (a port of the reference's torch-CPU train step, dense [B*S, N] logits)
                 timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only);
  elbo/parity  - ELBO terms of the last step and HIP-vs-oracle relative error on the baseline sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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


TIMER_GATHER, TIMER_ASSEMBLE = 1, 2   # include/pcvae.h: PCVAE_TIMER_*


def kernel_timer_run(fn, tag):
    """run fn() with the library's per-kernel timer on -> durations (ms) of the launches with this tag, in launch order"""
    import ctypes
    from pivotcvae_amd import _hip
    L = _hip.lib()
    _hip.check(L.pcvae_kernel_timer(1), "kernel_timer")
    try:
        fn()
        torch.cuda.synchronize()
        n = L.pcvae_kernel_timer_read(None, None, 0)
        ms, tags = (ctypes.c_float * max(n, 1))(), (ctypes.c_int * max(n, 1))()
        if L.pcvae_kernel_timer_read(ms, tags, n) < 0:
            raise RuntimeError("kernel_timer_read failed")
        return [ms[i] for i in range(n) if tags[i] == tag]
    finally:
        L.pcvae_kernel_timer(0)


def gather_roofline(model, cfg, device, tables=4):
    """K1 on its own: the (S+2)*B embedding rows of one step against the 8 TB/s HBM peak, caches cold (512 MB written
    before every measurement, > the 256 MB Infinity Cache).  `frac` = ONE launch, timed by HIP events attached to that dispatch
    (the kernel's own begin / end timestamps; profiles/ holds the rocprofv3 kernel trace + FETCH / WRITE counters of the same
    kernel).  Beside it: one launch between a hipEventRecord pair (carries the pair's own ~2.4 us: an empty kernel measures
    6.0 us event-to-event and 3.6 us in rocprofv3's trace, tools/gather_probe.hip) and `tables` launches back to back."""
    from pivotcvae_amd import ops
    N, S, D, B = cfg["N"], cfg["S"], cfg["D"], cfg["B"]
    g = torch.Generator(device=device).manual_seed(3)
    n_idx = B * (S + 2)
    tabs = [model.docEmbed.weight] + [torch.rand(N, D, device=device, generator=g) for _ in range(tables - 1)]
    idxs = [torch.randint(0, N, (n_idx,), device=device, generator=g) for _ in range(tables)]
    outs = [torch.empty(n_idx, D, device=device) for _ in range(tables)]
    flush = torch.empty(128 * 1024 * 1024, device=device)  # 512 MB > the 256 MB Infinity Cache
    nbytes = n_idx * (2 * D * 4 + 8)  # rows read + rows written + int64 indices
    ms = {}
    for mode, k in (("single", 1), ("back_to_back", tables)):
        ts = []
        for it in range(13):
            flush.fill_(float(it))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(k):
                ops.gather_rows(tabs[j], idxs[j], out=outs[j])
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) / k)
        ms[mode] = sum(ts) / len(ts)
    # the kernel's OWN duration: HIP events attached to the dispatch (hipExtLaunchKernelGGL start / stop events = the timestamps
    # rocprofv3's kernel trace shows), one launch at a time on a cold cache
    kt = []
    for it in range(13):
        flush.fill_(float(it))
        torch.cuda.synchronize()
        d = kernel_timer_run(lambda: ops.gather_rows(tabs[it % tables], idxs[it % tables], out=outs[it % tables]), TIMER_GATHER)
        if it >= 3:
            kt += d
    tk_mean = sum(kt) / len(kt)
    tk = sorted(kt)[len(kt) // 2]   # the MEDIAN of the ten cold launches (one launch each): robust against the odd 23 us outlier
    t1, tb = ms["single"], ms["back_to_back"]
    bw = lambda t_ms: nbytes / (t_ms * 1e-3) / 1e9
    from pivotcvae_amd import _hip
    gname = {0: "gather_rows_scalar_kernel", 1: "gather_rows_vec4_kernel", 2: "gather_rows_coal_kernel"}[
        _hip.lib().pcvae_gather_rows_variant(D, 1, D)]   # the kernel this width launches, as rocprofv3's trace names it
    return {"kernel": gname, "bound": "hbm", "achieved": bw(tk), "peak": 8000.0,
            "unit": "GB/s", "frac": bw(tk) / 8000.0, "bytes_per_launch": nbytes, "us_per_launch": tk * 1e3,
            "rows": n_idx, "timed_over": "ONE launch at a time, cold caches, HIP events attached to the dispatch (hipExtLaunchKernelGGL start / "
                                         "stop events: the kernel's own begin / end timestamps, as in rocprofv3's kernel trace); "
                                         "median of 10 such launches (rounds 1-3 reported the MEAN, kept as frac_of_mean; the median "
                                         "is robust against the odd 23 us outlier)",
            "us_per_launch_mean": tk_mean * 1e3, "us_per_launch_min": min(kt) * 1e3, "us_per_launch_max": max(kt) * 1e3,
            "frac_of_mean": bw(tk_mean) / 8000.0,
            "event_pair_around_one_launch": {"us_per_launch": t1 * 1e3, "achieved": bw(t1), "frac": bw(t1) / 8000.0,
                                             "note": "hipEventRecord pair around one launch: also times its own two marker packets (~2.4 us)"},
            "back_to_back": {"us_per_launch": tb * 1e3, "achieved": bw(tb), "frac": bw(tb) / 8000.0,
                             "note": f"{tables} launches on {tables} distinct cold tables between one event pair"},
            "achievable_hbm": {"GB/s": 6290.0, "frac_of_it": bw(tk) / 6290.0,
                               "note": "MI355X_MICROARCH.md: 6.29 TB/s measured for a float4 copy (79 % of the 8 TB/s spec)"},
            "cache": "cold (512 MB written before every measurement)",
            "rocprofv3_committed": committed_traffic("_r05_gather")}


ASSEMBLE_RESULT = {}   # filled by mlp_roofline (the same eager steps): the train step's fused gather kernel


def mlp_roofline(trainer, s, r, u, B, lo, steps=3, arithmetic="f32"):
    """MFMA utilisation of the MLP stacks (K3).  Every pass of a stack - the forward of encoder || prior, the forward of the
    slate-completion stack, and their two backward passes: runs of dependent GEMM launches with nothing between them - is
    bracketed with ONE pair of HIP events on the launch stream (ops.gemm_span), `steps` eager train steps;
    achieved = sum of 2*M*N*K over the launches / sum of the intervals, against the dense f32 MFMA peak (the MLPs compute in exact
    fp32: v_mfma_f32_32x32x2_f32).  The intervals include the gaps between a pass's launches and the ~2.4 us an event pair costs, so
    the figure is a lower bound of what the kernel durations in rocprofv3's trace give."""
    from pivotcvae_amd import ops
    ev = []

    def begin():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0, e1

    def end(tok, flops, launches):
        tok[1].record()
        ev.append((flops, launches, tok[0], tok[1]))

    was = trainer.capture_graph
    trainer.capture_graph = False
    trainer.step(s, r, u, global_batch=B, row_offset=lo)
    asm_ev = []
    ops.GEMM_TIMING = (begin, end)
    ops.ASSEMBLE_TIMING = (begin, lambda tok, nbytes: (tok[1].record(), asm_ev.append((nbytes, tok[0], tok[1]))))
    try:
        for _ in range(steps):
            trainer.step(s, r, u, global_batch=B, row_offset=lo)
        torch.cuda.synchronize()
        ops.GEMM_TIMING = None
        ops.ASSEMBLE_TIMING = None
        # the same kernel inside `steps` more eager steps, by the events attached to its own dispatch
        asm_kernel_ms = kernel_timer_run(lambda: [trainer.step(s, r, u, global_batch=B, row_offset=lo) for _ in range(steps)],
                                         TIMER_ASSEMBLE)
    finally:
        ops.GEMM_TIMING = None
        ops.ASSEMBLE_TIMING = None
        trainer.capture_graph = was
    if asm_ev:   # the train step's own gather (item / user / pivot rows + one-hot click count + the concatenations, ONE launch)
        a_ms = sum(a.elapsed_time(b) for _, a, b in asm_ev) / len(asm_ev)
        ASSEMBLE_RESULT.clear()
        k_ms = sum(asm_kernel_ms) / len(asm_kernel_ms) if asm_kernel_ms else a_ms
        ASSEMBLE_RESULT.update({"kernel": "assemble_inputs_vec_kernel", "bound": "hbm", "bytes_per_launch": asm_ev[0][0],
                                "us_per_launch": k_ms * 1e3, "achieved": asm_ev[0][0] / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                                "unit": "GB/s", "frac": asm_ev[0][0] / (k_ms * 1e-3) / 8e12,
                                "timed_over": "HIP events attached to the kernel's own dispatch, inside eager train steps",
                                "event_pair_around_the_launch": {"us_per_launch": a_ms * 1e3, "frac": asm_ev[0][0] / (a_ms * 1e-3) / 8e12},
                                "note": "S item rows + the user row read once, written into the encoder / prior / slate-completion inputs "
                                        "and slot 0 of rx together with the one-hot click count"})
    ms = sum(a.elapsed_time(b) for _, _, a, b in ev)
    flops = sum(f for f, _, _, _ in ev)
    tf = flops / (ms * 1e-3) / 1e12
    out = {"kernel": "gemm_group_kernel (all MLP GEMMs of a train step - fwd, input-grad, weight-grad - as grouped launches of independent layers)",
           "bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS["f32"],
           "launches_per_step": sum(n for _, n, _, _ in ev) // steps, "timed_intervals_per_step": len(ev) // steps,
           "ms_per_step": ms / steps, "flops_per_step": flops / steps,
           "timed_over": "one HIP event pair per stack pass (fwd enc||prior, fwd scm, bwd scm, bwd enc||prior), launch gaps included",
           "note": "PSM stack skipped in gt training (it never receives a gradient: SURVEY 0.7); the slate-completion stack's bottom "
                   "input gradient covers the z columns only (the rest of its input comes from frozen tables)"}
    if arithmetic == "bf16x3":
        # priced against the pipe it runs on: three bf16 MFMAs per algorithmic multiply-add against the dense bf16 peak (never > 1);
        # the algorithmic rate against the f32 MFMA peak stays beside it as a comparison with the exact-f32 GEMMs, not as a roofline
        out.update({"peak": PEAK_TFLOPS["bf16"], "frac": 3.0 * tf / PEAK_TFLOPS["bf16"], "mfmas_per_multiply_add": 3,
                    "frac_definition": "3 x achieved (MFMAs issued) / dense bf16 peak",
                    "algorithmic_vs_f32_mfma_peak": tf / PEAK_TFLOPS["f32"]})
    return out


def eval_throughput(model, cfg, device, bs=1024, trials=2):
    """Config 5: the in-loop evaluation of train_generative.py:169-195 (sample users -> 5 contexts x greedy slates ->
    click model -> min/mean/max expected clicks), `trials` trials of `bs` users on the device."""
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP
    from pivotcvae_amd.train_generative import recommendation_test
    S, D = cfg["S"], cfg["D"]
    torch.manual_seed(5)
    resp = UserResponseModel_MLP(8, N_USER - 1, D, S, [(S + 1) * D, 256, 256, S], device, False)
    resp.docEmbed = model.docEmbed  # same catalog (the click model's own table is a 10 GB duplicate at N = 10M)
    resp.maxItemId = cfg["N"] - 1
    resp = resp.to(device)
    recommendation_test(model, resp, bs, n_test_trial=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = recommendation_test(model, resp, bs, n_test_trial=trials)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_slates = trials * 5 * bs
    # what the time goes to: the S + 1 catalog argmaxes per slate (pivot + S slots; bf16 screening + exact rescoring for D in
    # (64, 128, 256)): their algorithmic 2 (S + 1) N D flops per slate against the dense peak of the pipe they run on
    flops = 2.0 * (S + 1) * cfg["N"] * D * n_slates
    peak = PEAK_TFLOPS["bf16"] if D in (64, 128, 256) else PEAK_TFLOPS["f32"]
    return {"value": n_slates / dt, "unit": "slates/s (generated AND scored)", "seconds": dt, "trials": trials, "users_per_trial": bs,
            "slates_generated": n_slates, "slates_scored_by_the_click_model": n_slates,
            "argmax_algorithmic_TFLOPs": flops / dt / 1e12, "argmax_frac_of_peak": flops / dt / 1e12 / peak,
            "reference": "train_generative.py:169-195 (5 contexts x trials; sample_users -> recommend -> resp_model -> sigmoid sums)",
            "expected_clicks_min_mean_max_per_context": [[round(float(v), 4) for v in row] for row in stats.cpu()]}


def pretrain_env_block(cfg, device, steps=5):
    """Training the click model (pretrain_env.py:25-139: gather + whole-vector normalisation + ReLU MLP + BCE of the sigmoid +
    backward incl. the embedding scatter-add + Adam with weight decay over ALL parameters, the item and user tables included) at this
    config's shape, one resident batch.  With an N x D table among the parameters the step is the optimiser's stream over it:
    zero-grad (1 write) + Adam (p, g, m, v read, p, m, v written) = 8 x 4 bytes per parameter against the HBM peak."""
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP
    from pivotcvae_amd.pretrain_env import ResponseTrainer
    N, S, D, B = cfg["N"], cfg["S"], cfg["D"], cfg["B"]
    torch.manual_seed(6)
    rm = UserResponseModel_MLP(8, N_USER - 1, D, S, [(S + 1) * D, 256, 256, S], "cpu", False)
    a = (2.0 / D) ** 0.5
    rm.docEmbed = torch.nn.Embedding(N, D, device=device)    # built on the device (a 10 GB host tensor is not needed for timing)
    rm.docEmbed.weight.data.uniform_(-a, a)
    rm.maxItemId = N - 1
    rm = rm.to(device)
    rm.device = device
    tr = ResponseTrainer(rm, lr=1e-3, decay=1e-5)
    s, r, u = synthetic_batch(cfg, B, device, seed=21)
    for _ in range(2):
        tr.step(s, u, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(s, u, r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tr.opt.zero_grad()
    tr.opt.step()
    e1.record()
    torch.cuda.synchronize()
    opt_ms = e0.elapsed_time(e1)
    n_par = tr.opt.flat.numel()
    nbytes = 8.0 * 4 * n_par
    t0 = time.perf_counter()
    for _ in range(steps):
        vl = tr.validation_loss(s, u, r)
    torch.cuda.synchronize()
    dv = (time.perf_counter() - t0) / steps
    out = {"value": 1.0 / dt, "unit": "steps/s", "slates_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B, "loss": float(loss),
           "parameters": n_par, "of_which_item_table": N * D,
           "dominant_kernel": {"kernel": "zero_kernel + adam_kernel over the flat buffer (the item table is a trained parameter "
                                         "with weight decay: pretrain_env.py:59)", "bound": "hbm", "ms_per_step": opt_ms,
                               "bytes_per_step": nbytes, "achieved": nbytes / (opt_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": nbytes / (opt_ms * 1e-3) / 8e12, "share_of_step": opt_ms / (dt * 1e3)},
           "validation": {"ms_per_batch": dv * 1e3, "slates_per_s": B / dv, "loss": float(vl),
                          "note": "no-grad forward + BCE (pretrain_env.py:96-108)"},
           "reference": "pretrain_env.py:76-92 (zero_grad, forward, BCELoss(sigmoid), backward, Adam.step with weight_decay)"}
    del tr, rm
    torch.cuda.empty_cache()
    return out


def validation_block(model, trainer, cfg, s, r, u, steps=3):
    """The epoch loop's validation pass (train_generative.py:151-165: get_gen_loss under no_grad at n_neg = the dataset's candidate
    count, default 1000), forward only: mask-train mode (sparse kept-rows kernel) and candidate mode (fused candidate kernel)."""
    from pivotcvae_amd import ops
    B = s.shape[0]
    out = {}
    for name, kw in (("mask_train_n_neg_1000", dict(n_neg=1000)), ("candidates_1000", dict(candidates=1000))):
        ev = []

        def begin():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            return e0, e1

        with torch.no_grad():
            model.loss(s, r, u, BETA, mask_seed=0x5641, **kw)
            torch.cuda.synchronize()
            ops.CATALOG_CE_TIMING = (begin, lambda p: (p[1].record(), ev.append(p)))
            t0 = time.perf_counter()
            for _ in range(steps):
                loss, rec, kld = model.loss(s, r, u, BETA, mask_seed=0x5641, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            ops.CATALOG_CE_TIMING = None
        k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        out[name] = {"ms_per_batch": dt * 1e3, "slates_per_s": B / dt, "loss": float(loss), "recLoss": float(rec),
                     "dominant_kernel": {"kernel": "catalog_ce_sparse_kernel<%d, false>" % cfg["D"] if "n_neg" in kw
                                         else "candidate_ce_kernel<%d, false>" % cfg["D"], "ms_per_launch": k_ms,
                                         "share_of_batch": k_ms / (dt * 1e3)}}
    out["reference"] = "train_generative.py:151-165 (model.eval(); no_grad; get_gen_loss(..., n_neg = valset.nCandidate))"
    return out


def generate_throughput(model, cfg, device, iters=3):
    """Greedy slate generation (recommend(return_item=True)): prior MLP -> z -> PSM -> catalog argmax (pivot) -> SCM ->
    catalog argmax (S slots).  Ids are always the exact fp32 ones (bit-exact against the reference arithmetic); for
    D in (64, 128, 256) the argmax runs as bf16 MFMA screening + exact fp32 rescoring of the candidates, timed here next to the
    plain f32-MFMA kernel, and both id sets are compared."""
    from pivotcvae_amd import ops
    B, S = cfg["B"], cfg["S"]
    g = torch.Generator(device=device).manual_seed(7)
    u = torch.randint(0, N_USER, (B, 1), device=device, generator=g)
    ctx = (torch.rand(B, S, device=device, generator=g) < 0.5).float()
    eps = torch.randn(B, Z, device=device, generator=g)  # same latent draw for both routes so that the ids can be compared
    flops = 2.0 * (S + 1) * cfg["N"] * cfg["D"] * B  # pivot argmax + S slot argmaxes (F_generate)
    res, ids = {}, {}
    saved = ops.SCREENED_MIN_ITEMS
    try:
        for name, min_items in (("screened", saved), ("f32", 1 << 62)):
            ops.SCREENED_MIN_ITEMS = min_items
            with torch.no_grad():
                model.recommend(ctx, u, return_item=True, eps=eps)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    items, _ = model.recommend(ctx, u, return_item=True, eps=eps)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
            ids[name] = items
            res[name] = {"slates_per_s": B / dt, "ms_per_batch": dt * 1e3, "algorithmic_TFLOPs": flops / dt / 1e12}
    finally:
        ops.SCREENED_MIN_ITEMS = saved
    # how many of the generated ids are NOT decided beyond fp32 rounding (top-2 margin <= 1e-5: the qualification SURVEY 7 attaches
    # to "bit-exact ids"), and do the ids agree with an fp64 argmax on the rows that are: a sample of rows, scores by torch in fp64
    # on the device, chunked over the catalog (measurement only)
    margin = None
    if cfg["N"] * cfg["D"] <= 2.6e8:
        with torch.no_grad():
            rx, _ = model.recommend(ctx[:64], u[:64], return_item=False, eps=eps[:64])
        rxs = rx.reshape(-1, cfg["D"]).double()
        E = model.docEmbed.weight.detach()
        top = torch.full((rxs.shape[0], 2), -float("inf"), dtype=torch.float64, device=device)
        arg = torch.zeros(rxs.shape[0], dtype=torch.int64, device=device)
        step = max(1, int(2.5e8 // rxs.shape[0]))
        for c0 in range(0, cfg["N"], step):
            sc = rxs @ E[c0:c0 + step].double().t()
            v, i = torch.topk(sc, min(2, sc.shape[1]), dim=1)
            better = v[:, 0] > top[:, 0]
            arg = torch.where(better, i[:, 0] + c0, arg)
            top = torch.topk(torch.cat([top, v], 1), 2, dim=1)[0]
        safe = (top[:, 0] - top[:, 1]) > 1e-5
        got = ids["screened"][:rxs.shape[0]]
        margin = {"rows_checked": int(rxs.shape[0]), "rows_with_top2_margin_below_1e-5": int((~safe).sum()),
                  "ids_equal_fp64_argmax_on_the_safe_rows": bool(torch.equal(got[safe], arg[safe])),
                  "ids_equal_fp64_argmax_on_all_rows": bool(torch.equal(got, arg))}
    screened = cfg["D"] in ops.BF16_DIMS and cfg["N"] >= saved
    best = res["screened"]
    # the screening pass does the algorithmic 2*R*N*D flops once over the whole catalog (+1/16 for the prefix pass)
    peak = PEAK_TFLOPS["bf16"] if screened else PEAK_TFLOPS["f32"]
    return {"value": best["slates_per_s"], "unit": "slates/s", "ms_per_batch": best["ms_per_batch"],
            "arithmetic": ("bf16 MFMA screening + exact fp32 rescoring (bit-exact greedy ids)" if screened
                           else "f32 MFMA (bit-exact greedy ids)"),
            "achieved_TFLOPs": best["algorithmic_TFLOPs"], "peak_TFLOPs": peak, "frac": best["algorithmic_TFLOPs"] / peak,
            "f32_kernel": res["f32"], "ids_identical_to_f32_kernel": bool(torch.equal(ids["screened"], ids["f32"])),
            "margin_safety": margin}


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
            "sample": f"oracle/pivotcvae_oracle.py train step in candidate mode (gathered rows [{Bs}, {S}, {n_candidate}, {D}] + bmm + CE + "
                      f"KL + backward + Adam: {dt:.3f} s/step) + the candidate draw the reference's way ({t_draw:.3f} s per batch), "
                      f"B={Bs} slates of the same workload, {steps} steps"}
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
    ranks; HIP events on the launch stream around the dominant kernel inside those steps"""

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

        pivot_events = []

        def pivot_end(pair):
            pair[1].record()
            pivot_events.append(pair)

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
        self.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, rec, kld = tr.step(s, r, u, global_batch=self.B, row_offset=self.lo)
            if trace is not None:
                trace.append((loss, rec, kld))
        self.sync_all()
        dt = time.perf_counter() - t0
        if trace is not None:
            for i, t in enumerate(trace):
                print(f"[elbo trace] step {i}: " + " ".join(f"{float(v):.6f}" for v in t), file=sys.stderr, flush=True)
        ops.CATALOG_CE_TIMING = None
        ops.PIVOT_TIMING = None
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
        return dict(dt=dt, steps=steps, kern_ms=kern_ms, pivot_ms=pivot_ms, graphed=graphed, elbo=(loss, rec, kld))


# what the catalog contraction computes in, per --dtype: (json dtype, MFMA peak it is priced against, MFMAs issued per
# algorithmic MAC).  bf16x3 = hi/lo bf16 split of BOTH operands, three bf16 MFMAs per product with fp32 accumulation: fp32-
# equivalent results (tests/test_hip_x3.py: same tolerances as the f32 kernel) on the bf16 pipe.
ARITH = {"f32": ("f32", PEAK_TFLOPS["f32"], 1), "bf16": ("bf16", PEAK_TFLOPS["bf16"], 1),
         "bf16x3": ("bf16x3", PEAK_TFLOPS["bf16"], 3), "bf16x6": ("bf16x6 (24-bit operands = the fp32 values, fp32-exact products, fp32 accumulate)", PEAK_TFLOPS["bf16"], 6)}


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
    _, peak, mult = ARITH[dtype]
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


def pivot_rules_block(cfg, device, dtype, mlp, gt_pi_ms):
    """The paper's variants (models/pivotcvae.py:321-455, settings.py:36-42) at this config, one model each from the registry:
    train step (3 timed steps after 2 warm-up, eager) with the pivot kernel's own time, and generation (recommend(return_item))
    for the sampled inference rule."""
    from pivotcvae_amd.train_generative import Trainer
    out = {"gt_pi_ms_per_step": gt_pi_ms}
    B, S = cfg["B"], cfg["S"]
    for key in ("pivotcvae_sgt_pi", "pivotcvae_spt_pi", "pivotcvae_pt_pi", "pivotcvae_gt_spi"):
        c2 = dict(cfg, model=key)
        m, _ = build_model(c2, device, dtype)
        m.set_mlp_precision(mlp)
        blk = {"train_rule": m.TRAIN_RULE, "infer_rule": m.INFER_RULE}
        if m.TRAIN_RULE != "gt":
            tr = Trainer(m, lr=LR, beta=BETA, capture_graph=False)
            s, r, u = synthetic_batch(c2, B, device)
            v = StepTimer(tr, (s, r, u), B, 0, False, device).run(3, 2)
            ms = v["dt"] / v["steps"] * 1e3
            blk["train"] = {"value": B / (ms * 1e-3), "unit": "slates/s", "ms_per_step": ms, "vs_gt_pi_step": ms / gt_pi_ms,
                            "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                            "pivot_kernel": pivot_block(c2, B, v["pivot_ms"], ms, m.TRAIN_RULE) if v["pivot_ms"] else None}
            del tr
        if m.INFER_RULE == "spi":
            g = torch.Generator(device=device).manual_seed(7)
            u = torch.randint(0, N_USER, (B, 1), device=device, generator=g)
            ctx = (torch.rand(B, S, device=device, generator=g) < 0.5).float()
            ev = []
            with torch.no_grad():
                m.recommend(ctx, u, return_item=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    m.recommend(ctx, u, return_item=True)
                torch.cuda.synchronize()
                dtg = (time.perf_counter() - t0) / 3
            blk["generate"] = {"value": B / dtg, "unit": "slates/s", "ms_per_batch": dtg * 1e3,
                               "note": "pivot by Categorical(sigmoid(scores)) (rejection sampler), the S slots by exact greedy argmax"}
        out[key] = blk
        del m
        torch.cuda.empty_cache()
    return out


def committed_traffic(key):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (PMC counters cannot be read from inside the run)"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get(key)


X3_ARITHMETIC = ("bf16x3 - a stated-tolerance fast path, NARROWER than the reference's fp32: operands as bf16 hi + lo (16-bit mantissa), "
                 "3 bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, lo*lo dropped: 2^-18 relative per product), fp32 accumulate; target "
                 "logit / target row in exact fp32; lse / nll within 2e-6, dx within 2e-5 of its scale vs the fp32 oracle "
                 "(tests/test_hip_x3.py); row blocks over the Cauchy-Schwarz logit bound run the exact f32 kernel")


def arithmetic_error_vs_fp64(model, cfg, r, u, device, rows=320):
    """Error of the catalog kernels against fp64 on rows the MODEL itself produces (rx of `recommend` for the first slates of the
    batch) over the whole table: max |lse - lse64| and max |dx - dx64| / max |dx64| per arithmetic.  fp64 softmax by torch on the
    device, chunked over the catalog (an independent path; measurement only, outside every timed region)."""
    from pivotcvae_amd import ops
    from pivotcvae_amd._hip import PREC_NAMES
    N, S, D = cfg["N"], cfg["S"], cfg["D"]
    nb = max(1, rows // S)
    with torch.no_grad():
        rx = model.recommend(r[:nb], u[:nb])[0].reshape(-1, D).contiguous()
    R = rx.shape[0]
    E = model.docEmbed.weight.detach()
    tgt = torch.randint(0, N, (R,), device=device, generator=torch.Generator(device=device).manual_seed(7))
    m = torch.full((R,), -float("inf"), device=device, dtype=torch.float64)
    ssum = torch.zeros(R, device=device, dtype=torch.float64)
    num = torch.zeros(R, D, device=device, dtype=torch.float64)
    step = max(1, min(N, int(2.5e8 // max(R, 1))))
    for c0 in range(0, N, step):
        Ec = E[c0:c0 + step].double()
        lg = rx.double() @ Ec.t()
        mn = torch.maximum(m, lg.max(1)[0])
        sc = torch.exp(m - mn)
        pe = torch.exp(lg - mn[:, None])
        ssum = ssum * sc + pe.sum(1)
        num = num * sc[:, None] + pe @ Ec
        m = mn
    lse64 = m + torch.log(ssum)
    dx64 = num / ssum[:, None] - E[tgt].double()
    out = {"rows": R, "row_source": "rx of model.recommend on the batch's first slates (real model outputs)", "items": N,
           "max_abs_lse64": float(lse64.abs().max()), "max_row_norm": float(rx.norm(dim=1).max())}
    table = model.catalog_table()
    for name in ("f32", "bf16x6", "bf16x3", "bf16"):
        if ops.effective_precision(PREC_NAMES[name], D) != PREC_NAMES[name]:
            continue
        _, lse, dx = ops.catalog_ce_raw(rx, table, tgt, prec=PREC_NAMES[name])
        el = lse.double() - lse64
        out[name] = {"lse_max_abs_err": float(el.abs().max()), "lse_rms_err": float(el.pow(2).mean().sqrt()),
                     "lse_mean_err": float(el.mean()),
                     "dx_max_err_over_scale": float((dx.double() - dx64).abs().max() / dx64.abs().max())}
    return out


X6_ARITHMETIC = ("bf16x6 - the reference's fp32 arithmetic on the bf16 matrix cores: every fp32 operand (table rows, rx rows, softmax "
                 "numerators) as THREE bf16 components whose sum is the fp32 value exactly (3 x 8 = 24 significand bits), 6 bf16 MFMAs per "
                 "product (c0c0, c0c1, c1c0, c1c1, c0c2, c2c0; the dropped c1c2, c2c1, c2c2 are <= 2^-25 relative: below the rounding of "
                 "an fp32 product), every partial product exact, fp32 accumulate; target logit / target row in exact fp32.  Against fp64 "
                 "its error is that of the exact f32-MFMA kernel on the same inputs (tests/test_hip_x6.py: err <= 2 x the f32 kernel's "
                 "+ 1 ulp on every shape, <= 4 x on cancelling / large-norm / dominant-logit rows, lse within one fp32 ulp of fp64; half "
                 "the f32 kernel's test tolerances against the fp32 oracle; `arithmetic_error_vs_fp64` measures it live); row blocks over the Cauchy-Schwarz logit bound run the exact f32 kernel.  `variants.f32` is the same "
                 "workload on v_mfma_f32_32x32x2_f32")


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
                         "named blocks under `variants`")
    ap.add_argument("--mlp", default=None, choices=["f32", "bf16x3"],
                    help="arithmetic of the MLP GEMMs of the train step.  Default: bf16x3 where the catalog contraction runs in bf16x3 "
                         "or bf16 (the whole step then computes on the bf16 matrix cores), exact f32 MFMA with --dtype f32 / bf16x6")
    ap.add_argument("--n_neg", type=int, default=None, help="default: N (full-catalog softmax)")
    ap.add_argument("--n_candidate", type=int, default=None,
                    help="time the reference's DEFAULT training mode instead (no --mask_train: candidate sets of this many ids per "
                         "slot, data_loader.py:46-58, train_generative.py:52-57) - the fused candidate kernel")
    ap.add_argument("--model", default=None,
                    help="registry key of the model (any of PIVOTCVAE_MODELS: pivotcvae_{gt,pt,spt,sgt}_{pi,spi}); default: the "
                         "config's (pivotcvae_gt_pi; config 1: listcvae)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the mlp_roofline / gather_roofline / generate / eval blocks")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` blocks (other arithmetics, n_neg = 1000)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", action="store_true", help="replay a hipGraph at any batch size (default: only when the "
                                                         "per-rank batch is <= 4096 slates, where launches matter)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="override the config's global batch (e.g. 1024 on one GPU = the per-rank load of the 8-GPU run)")
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
        # every fp32 operand carried exactly as three bf16 components; `variants.f32` is in the same line
        args.dtype = {"3": "bf16", "5": "bf16"}.get(args.config, "bf16x6" if D == 128 else "f32")
    if args.dtype == "bf16x3" and ops.x3_width(D) is None:
        raise SystemExit(f"bf16x3 exists for D <= {ops.X3_MAX_PADDED}")
    if args.dtype == "bf16x6" and ops.x6_width(D) is None:
        raise SystemExit("bf16x6 exists for D <= 128")
    if args.mlp is None:
        args.mlp = "bf16x3" if args.dtype in ("bf16x3", "bf16") else "f32"
    model, st = build_model(cfg, device, args.dtype)
    model.set_mlp_precision(args.mlp)
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
    # the gather kernels read bf16 rows where the config's stated arithmetic is bf16 (configs 3 / 5), the fp32 table otherwise
    bf16_rows = args.dtype == "bf16" and D in ops.BF16_DIMS
    rows_dtype = "bf16 rows, fp32 accumulate" if bf16_rows else "f32"
    if cand_mode:
        roof = candidate_roofline(R_local, N, D, args.n_candidate, kern_ms,
                                  committed_traffic(f"config{args.config}_cand{args.n_candidate}_gpus{world}"), bf16_rows)
        sparse = True
    else:
        roof = roofline_block(kernel_name(R_local, N, D, args.dtype) if not sparse else "catalog_ce_sparse_kernel",
                              R_local, N, D, args.dtype, kern_ms, sparse_kept=(args.n_neg + 1) if sparse else None,
                              traffic=committed_traffic(f"config{args.config}_nneg{args.n_neg}_gpus{world}") if sparse else None,
                              bf16_rows=bf16_rows)
    if not sparse:
        roof["kernel"] += " (events also span its row-bound prologue and merge kernels, <1% together)"
        roof["traffic"] = committed_traffic(f"config{args.config}_{args.dtype}_gpus{world}")
    roof["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of this kernel, committed; not measured in this run)"
    roof["timed_over"] = (f"{args.steps} eager steps right after the timed graph-replayed steps (HIP events cannot be "
                          "recorded inside a hipGraph)") if graphed else "the timed steps"

    out = {
        "metric": "slates/sec + ELBO, N=1M catalog K=10 B=8192" if args.config == "4" and not args.global_batch
                  else f"slates/sec config {args.config}",
        "value": B * args.steps / dt, "unit": "slates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": rows_dtype if (cand_mode or sparse) else ARITH[args.dtype][0], "data": "synthetic",
        "config": {"workload": f"{'ListCVAE' if cfg.get('model') == 'listcvae' else 'PivotCVAE ' + cfg.get('model', 'pivotcvae_gt_pi')[10:]} train step (fwd+bwd+Adam), catalog N={N} slate K={S} emb D={D} "
                               f"global batch B={B}, " + (f"candidate sets of {args.n_candidate} ids per slot drawn in-kernel (the reference's default mode: no --mask_train)"
                                                          if cand_mode else "full-catalog softmax" + ("" if args.n_neg is None else f" n_neg={args.n_neg}")),
                   "model": cfg.get("model", "pivotcvae_gt_pi"),
                   "global_batch": B, "per_gpu_batch": B // world, "parallelism": f"dp{world}",
                   "rccl_ranks": dist.get_world_size() if use_dist else 1,
                   **({"rehearsal": "all ranks on ONE GPU, gloo collectives: checks the N > 1 path, not its speed"} if rehearsal else {}),
                   "catalog_arithmetic": ("gather kernel: " + ("rows of the bf16 table widened exactly, " if bf16_rows else "rows of the fp32 table, ")
                                          + "fp32 fmaf dot products, fp32 online softmax (the full-catalog MFMA kernels are not on this path)")
                   if (cand_mode or sparse) else {"bf16x3": X3_ARITHMETIC, "bf16x6": X6_ARITHMETIC}.get(args.dtype, args.dtype),
                   "mlp_arithmetic": args.mlp if args.mlp == "f32" else
                   "bf16x3 in the train step's 64 x 64-tile GEMM launches (operands split into bf16 hi + lo in registers, 3 bf16 MFMAs per "
                   "product, fp32 accumulate: gradients within 1e-4 of each tensor's scale of the fp32 reference, ELBO ~1e-6; "
                   "tests/test_hip_stated_goldens.py); generation: exact f32",
                   "launch": "hipGraph replay (zero-grad+fwd+bwd) + eager all-reduce + Adam" if graphed else "eager"},
        "elbo": {"loss": loss.item(), "recLoss": rec.item(), "KLD": kld.item()},
        "roofline": roof,
    }
    if res.get("pivot_ms") is not None:
        out["pivot_kernel"] = pivot_block(cfg, B // world, res["pivot_ms"], dt / args.steps * 1e3, getattr(model, "TRAIN_RULE", "gt"))
    single = rank == 0 and world == 1
    if single and not args.no_variants:
        # the same workload in the other arithmetics and in the reference's default masked mode (n_neg = 1000), each with its
        # own timed region (2 warm-up + 3 steps) and roofline; the headline above is never taken from here
        variants = {}
        was_graph = trainer.capture_graph
        trainer.capture_graph = False
        for dt_name in ("f32", "bf16x6", "bf16x3", "bf16"):
            if dt_name == args.dtype or (dt_name == "bf16x3" and ops.x3_width(D) is None) or \
                    (dt_name == "bf16x6" and (ops.x6_width(D) is None or D < 64)) or \
                    (dt_name == "bf16" and D not in ops.BF16_DIMS) or args.n_neg is not None or cand_mode:
                continue
            if dt_name == "f32" and 4.0 * R_local * N * D > 2e14:   # config 5 in exact f32: minutes per step
                continue
            model.set_catalog_precision(dt_name)
            v = timer.run(3, 2)
            variants[dt_name] = {"value": B * v["steps"] / v["dt"], "unit": "slates/s", "ms_per_step": v["dt"] / v["steps"] * 1e3,
                                 "dtype": ARITH[dt_name][0],
                                 "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                                 "roofline": roofline_block(kernel_name(R_local, N, D, dt_name), R_local, N, D, dt_name, v["kern_ms"])}
        model.set_catalog_precision(args.dtype)
        def light_variant(**mode):
            # its own Trainer on the same replica and optimiser: these steps are a few ms, so they replay as a hipGraph (round 5: the
            # kernels read their seed from a device word); the kernel itself is timed in eager steps right after (StepTimer)
            tr2 = Trainer(model, lr=LR, beta=BETA, capture_graph=not args.no_graph, resident_batch=True, optimizer=trainer.opt, **mode)
            v = StepTimer(tr2, (s, r, u), B, lo, use_dist, device).run(5, 2)
            v["launch"] = "hipGraph replay (zero-grad+fwd+bwd) + eager Adam" if v["graphed"] else "eager"
            return v

        if args.n_neg is None and N >= 100_000:
            v = light_variant(n_neg=1000)   # train_generative.py:44 default; in-kernel Philox keep set (sparse path: only kept rows are read)
            variants["n_neg_1000"] = {"value": B * v["steps"] / v["dt"], "unit": "slates/s", "ms_per_step": v["dt"] / v["steps"] * 1e3,
                                      "dtype": rows_dtype, "launch": v["launch"],
                                      "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                                      "roofline": roofline_block("catalog_ce_sparse_kernel", R_local, N, D, "f32", v["kern_ms"],
                                                                 sparse_kept=1001,
                                                                 traffic=committed_traffic(f"config{args.config}_nneg1000_gpus{world}"),
                                                                 bf16_rows=bf16_rows)}
        if args.n_neg is None and not cand_mode and N >= 100_000:
            # the reference's DEFAULT mode (train_generative.py:270-274: candidate sets unless --mask_train; my_utils.py:169
            # --nneg 1000): ONE fused launch per step draws the sets, gathers, scores, takes the CE and the gradient
            for cn in (1000, 50):
                v = light_variant(n_candidate=cn)
                variants[f"candidates_nneg{cn}"] = {
                    "value": B * v["steps"] / v["dt"], "unit": "slates/s", "ms_per_step": v["dt"] / v["steps"] * 1e3, "dtype": rows_dtype,
                    "launch": v["launch"],
                    "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                    "roofline": candidate_roofline(R_local, N, D, cn, v["kern_ms"],
                                                   committed_traffic(f"config{args.config}_cand{cn}_gpus{world}"), bf16_rows)}
        trainer.capture_graph = was_graph
        out["variants"] = variants
    if single and not args.no_variants and cfg.get("model") != "listcvae" and N * D <= 2.6e8:
        # what "the reference's arithmetic" means in numbers: every arithmetic of this line against fp64, measured live
        out["arithmetic_error_vs_fp64"] = arithmetic_error_vs_fp64(model, cfg, r, u, device)
    if single and not args.no_variants and not args.no_extras and cfg.get("model", "pivotcvae_gt_pi") == "pivotcvae_gt_pi" \
            and args.n_neg is None and not cand_mode and N >= 100_000:
        out["pivot_rules"] = pivot_rules_block(cfg, device, args.dtype, args.mlp, out["ms_per_step"])
    if single and not args.no_cpu_baseline and cand_mode and cfg.get("model", "pivotcvae_gt_pi") in ("pivotcvae_gt_pi", "listcvae"):
        out["cpu_baseline"], out["parity"] = cpu_baseline_candidates(model, st, cfg, args.n_candidate)
    if single and not args.no_cpu_baseline and cfg.get("model", "pivotcvae_gt_pi") in ("pivotcvae_gt_pi", "listcvae") and not cand_mode:
        base, parity = cpu_baseline_and_parity(model, st, cfg, args.dtype)
        out["cpu_baseline"] = base
        out["parity"] = parity
    if single and not args.no_extras:
        out["mlp_roofline"] = mlp_roofline(trainer, s, r, u, B, lo, arithmetic=args.mlp)
        out["mlp_roofline"]["arithmetic"] = args.mlp
        other = "f32" if args.mlp == "bf16x3" else "bf16x3"
        model.set_mlp_precision(other)
        asm_keep = dict(ASSEMBLE_RESULT)
        out["mlp_roofline_" + other] = dict(mlp_roofline(trainer, s, r, u, B, lo, arithmetic=other), arithmetic=other)
        ASSEMBLE_RESULT.clear()
        ASSEMBLE_RESULT.update(asm_keep)
        model.set_mlp_precision(args.mlp)
        out["gather_roofline"] = gather_roofline(model, cfg, device, tables=4 if N * D * 4 <= (1 << 30) else 2)
        if ASSEMBLE_RESULT:
            out["gather_roofline"]["train_step_kernel"] = dict(ASSEMBLE_RESULT)
        out["generate"] = generate_throughput(model, cfg, device)
        if N >= 100_000 and cfg.get("model", "pivotcvae_gt_pi") != "listcvae":
            # the other two phases of the reference's epoch loop (validation) and the click model's own training (pretrain_env)
            out["validation"] = validation_block(model, trainer, cfg, s, r, u)
            out["pretrain_env"] = pretrain_env_block(cfg, device)
        if args.config == "5":
            out["eval"] = eval_throughput(model, cfg, device)
    if use_dist:
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio; push it out first so that the JSON line is the LAST line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stderr.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
