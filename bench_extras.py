"""Side blocks of the benchmark - everything `bench.py` measures BESIDE its headline (round 6: moved out of bench.py so that the
file the driver runs is the headline path, and out of the ONE JSON line so that the line stays parseable):

  variants          the same workload in the other catalog arithmetics, mask-train n_neg = 1000, candidate sets (Cn = 1000 / 50)
  pivot_rules       one model per paper variant (sgt / spt / pt train steps, spi generation)
  mlp_roofline*     K3 (the MLP GEMMs of a train step) per arithmetic
  gather_roofline   K1 (the embedding gather) on its own + the train step's fused gather kernel
  generate          greedy slate generation (recommend(return_item=True))
  validation        the epoch loop's validation pass
  pretrain_env      the click model's own training step
  epoch             one epoch of train_on_dataset over 16 x B resident slates (candidate mode and mask-train), incl. validation
  eval              (config 5) the in-loop recommendation test
  arithmetic_error_vs_fp64

`run(ctx)` returns them as one dict; bench.py writes it (with the verbose headline) to `bench_extras.json` and copies a few
two-number summaries into the line.  A block that raises is recorded as {"error": ...} - it never costs the headline."""
import time
import traceback

import torch

from bench import (ARITH, BETA, LR, N_USER, PEAK_TFLOPS, Z, StepTimer, build_model, candidate_roofline, committed_traffic,
                   kernel_name, pivot_block, roofline_block, synthetic_batch)


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
            "rocprofv3_committed": committed_traffic("_r06_gather") or committed_traffic("_r05_gather")}



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
    mult = {"bf16x3": 3, "bf16x6": 6}.get(arithmetic)
    if mult:
        # priced against the pipe it runs on: `mult` bf16 MFMAs per algorithmic multiply-add against the dense bf16 peak (never > 1);
        # the algorithmic rate against the f32 MFMA peak stays beside it as a comparison with the exact-f32 GEMMs, not as a roofline
        out.update({"peak": PEAK_TFLOPS["bf16"], "frac": mult * tf / PEAK_TFLOPS["bf16"], "mfmas_per_multiply_add": mult,
                    "frac_definition": f"{mult} x achieved (MFMAs issued) / dense bf16 peak",
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
                items, _ = model.recommend(ctx, u, return_item=True, eps=eps)
                torch.cuda.synchronize()
                n_it = max(iters, min(200, int(0.05 / max(time.perf_counter() - t0, 1e-6))))   # ~50 ms of batches (a batch is 0.2 .. 40 ms)
                t0 = time.perf_counter()
                for _ in range(n_it):
                    items, _ = model.recommend(ctx, u, return_item=True, eps=eps)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n_it
            ids[name] = items
            res[name] = {"slates_per_s": B / dt, "ms_per_batch": dt * 1e3, "algorithmic_TFLOPs": flops / dt / 1e12, "launch": "eager",
                         "batches_timed": n_it}
        # the same chain (~33 launches) captured once as a hipGraph and replayed - what the in-loop evaluation does
        # (train_generative.recommendation_test(capture_graph=True)); it pays where the batch is launch-bound (configs 1-2)
        ops.SCREENED_MIN_ITEMS = saved
        try:
            with torch.no_grad():
                warm = {}
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), ops.workspace_holder(warm):
                    for _ in range(2):
                        model.recommend(ctx, u, return_item=True, eps=eps)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"), ops.workspace_holder(warm):
                    items_g, _ = model.recommend(ctx, u, return_item=True, eps=eps)
                gr.replay()
                torch.cuda.synchronize()
                n_it = res["screened"]["batches_timed"]
                t0 = time.perf_counter()
                for _ in range(n_it):
                    gr.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n_it
            res["graph"] = {"slates_per_s": B / dt, "ms_per_batch": dt * 1e3, "algorithmic_TFLOPs": flops / dt / 1e12,
                            "launch": "hipGraph replay", "batches_timed": n_it, "ids_identical_to_eager": bool(torch.equal(items_g, ids["screened"]))}
            del gr
        except Exception as e:   # a measurement beside the eager one: say why it is missing
            torch.cuda.synchronize()
            res["graph"] = {"error": f"{type(e).__name__}: {e}"[:200]}
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
    g_ok = "slates_per_s" in res.get("graph", {}) and res["graph"]["ids_identical_to_eager"]
    if g_ok and res["graph"]["slates_per_s"] > best["slates_per_s"]:
        best = res["graph"]   # `value` = the faster of the two launch forms, named in `launch`
    # the screening pass does the algorithmic 2*R*N*D flops once over the whole catalog (+1/16 for the prefix pass)
    peak = PEAK_TFLOPS["bf16"] if screened else PEAK_TFLOPS["f32"]
    return {"value": best["slates_per_s"], "unit": "slates/s", "ms_per_batch": best["ms_per_batch"],
            "arithmetic": ("bf16 MFMA screening + exact fp32 rescoring (bit-exact greedy ids)" if screened
                           else "f32 MFMA (bit-exact greedy ids)"),
            "achieved_TFLOPs": best["algorithmic_TFLOPs"], "peak_TFLOPs": peak, "frac": best["algorithmic_TFLOPs"] / peak,
            "launch": best["launch"], "eager": res["screened"], "graph_replay": res.get("graph"),
            "f32_kernel": res["f32"], "ids_identical_to_f32_kernel": bool(torch.equal(ids["screened"], ids["f32"])),
            "margin_safety": margin}



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



def _guard(out, name, fn):
    """run one side block; a failure is recorded, never raised (the headline line must still be printed)"""
    try:
        v = fn()
        if v is not None:
            out[name] = v
    except Exception as e:   # noqa: BLE001 - measurement code: record and go on
        traceback.print_exc()
        out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()


def variants_block(ctx):
    """the same workload in the other arithmetics and in the reference's other loss modes (n_neg = 1000; candidate sets), each with
    its own timed region (2 warm-up + 3 / 5 steps) and roofline; the headline is never taken from here"""
    from pivotcvae_amd import ops
    from pivotcvae_amd.train_generative import Trainer
    args, cfg, model, trainer, timer = ctx["args"], ctx["cfg"], ctx["model"], ctx["trainer"], ctx["timer"]
    s, r, u = ctx["batch"]
    lo, device, use_dist, R_local = ctx["lo"], ctx["device"], ctx["use_dist"], ctx["R_local"]
    bf16_rows, rows_dtype = ctx["bf16_rows"], ctx["rows_dtype"]
    N, D, B = cfg["N"], cfg["D"], cfg["B"]
    world = 1
    cand_mode = args.n_candidate is not None
    variants = {}
    was_graph = trainer.capture_graph
    trainer.capture_graph = False
    try:
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
                                 "dtype": ARITH[dt_name][0], "arithmetic": ARITH[dt_name][3],
                                 "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                                 "roofline": roofline_block(kernel_name(R_local, N, D, dt_name), R_local, N, D, dt_name, v["kern_ms"])}
    finally:
        model.set_catalog_precision(args.dtype)

    def light_variant(**mode):
        # its own Trainer on the same replica and optimiser: these steps are a few ms, so they replay as a hipGraph (the kernels read
        # their seed from a device word); the kernel itself is timed in eager steps right after (StepTimer)
        tr2 = Trainer(model, lr=LR, beta=BETA, capture_graph=not args.no_graph, resident_batch=True, optimizer=trainer.opt, **mode)
        v = StepTimer(tr2, (s, r, u), B, lo, use_dist, device).run(5, 2)
        v["launch"] = "hipGraph replay (zero-grad+fwd+bwd) + eager Adam" if v["graphed"] else "eager"
        return v

    try:
        if args.n_neg is None and N >= 100_000:
            v = light_variant(n_neg=1000)   # train_generative.py:44 default; in-kernel Philox keep set (sparse path: only kept rows are read)
            variants["n_neg_1000"] = {"value": B * v["steps"] / v["dt"], "unit": "slates/s", "ms_per_step": v["dt"] / v["steps"] * 1e3,
                                      "dtype": rows_dtype, "launch": v["launch"],
                                      "elbo": {k: t.item() for k, t in zip(("loss", "recLoss", "KLD"), v["elbo"])},
                                      "roofline": roofline_block("catalog_ce_sparse_kernel", R_local, N, D, "f32", v["kern_ms"],
                                                                 sparse_kept=1001,
                                                                 traffic=committed_traffic(None if args.global_batch else f"config{args.config}_nneg1000_gpus{world}"),
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
                                                   committed_traffic(None if args.global_batch else f"config{args.config}_cand{cn}_gpus{world}"), bf16_rows)}
    finally:
        trainer.capture_graph = was_graph
    return variants


def mlp_blocks(ctx, out):
    """K3 per arithmetic: the headline's first (it also fills the train step's gather kernel), then the others"""
    from pivotcvae_amd import ops
    args, model, trainer = ctx["args"], ctx["model"], ctx["trainer"]
    s, r, u = ctx["batch"]
    B, lo = ctx["cfg"]["B"], ctx["lo"]
    out["mlp_roofline"] = dict(mlp_roofline(trainer, s, r, u, B, lo, arithmetic=args.mlp), arithmetic=args.mlp)
    asm_keep = dict(ASSEMBLE_RESULT)
    try:
        for other in ops.MLP_PRECISIONS:
            if other == args.mlp:
                continue
            model.set_mlp_precision(other)
            out["mlp_roofline_" + other] = dict(mlp_roofline(trainer, s, r, u, B, lo, arithmetic=other), arithmetic=other)
    finally:
        model.set_mlp_precision(args.mlp)
        ASSEMBLE_RESULT.clear()
        ASSEMBLE_RESULT.update(asm_keep)


def run(ctx):
    """every side block this configuration has -> one dict (merged into the extras file by bench.py)"""
    args, cfg, model, trainer, device = ctx["args"], ctx["cfg"], ctx["model"], ctx["trainer"], ctx["device"]
    s, r, u = ctx["batch"]
    N, D = cfg["N"], cfg["D"]
    mname = cfg.get("model", "pivotcvae_gt_pi")
    cand_mode = args.n_candidate is not None
    out = {}
    if not args.no_variants:
        _guard(out, "variants", lambda: variants_block(ctx))
        if mname != "listcvae" and N * D <= 2.6e8:
            # what "the reference's arithmetic" means in numbers: every arithmetic of this run against fp64, measured live
            _guard(out, "arithmetic_error_vs_fp64", lambda: arithmetic_error_vs_fp64(model, cfg, r, u, device))
        if not args.no_extras and mname == "pivotcvae_gt_pi" and args.n_neg is None and not cand_mode and N >= 100_000:
            _guard(out, "pivot_rules", lambda: pivot_rules_block(cfg, device, args.dtype, args.mlp, ctx["headline_ms"]))
    if not args.no_extras:
        _guard(out, "mlp_roofline", lambda: mlp_blocks(ctx, out))

        def gather():
            g = gather_roofline(model, cfg, device, tables=4 if N * D * 4 <= (1 << 30) else 2)
            if ASSEMBLE_RESULT:
                g["train_step_kernel"] = dict(ASSEMBLE_RESULT)
            return g
        _guard(out, "gather_roofline", gather)
        _guard(out, "generate", lambda: generate_throughput(model, cfg, device))
        if N >= 100_000 and mname != "listcvae":
            # the other two phases of the reference's epoch loop (validation) and the click model's own training (pretrain_env)
            _guard(out, "validation", lambda: validation_block(model, trainer, cfg, s, r, u))
            _guard(out, "pretrain_env", lambda: pretrain_env_block(cfg, device))
            _guard(out, "epoch", lambda: epoch_block(ctx))
        if args.config == "5":
            _guard(out, "eval", lambda: eval_throughput(model, cfg, device))
    return out


class _NullLogger:
    def log(self, msg):
        pass


def epoch_block(ctx, n_batches=16, val_batches=2):
    """One EPOCH of the loop the steps live in (train_generative.py:120-165 -> pivotcvae_amd.train_generative.train_on_dataset):
    `n_batches` x B training slates resident in HBM, a fresh device-side permutation, `tr_s[mine]` index ops per batch, the per-epoch
    host reads, then the validation pass over `val_batches` x B slates - in the reference's default mode (candidate sets, Cn = 1000)
    and in mask-train (n_neg = 1000).  Against it: the same Trainer stepping ONE resident batch (what `variants` times).
    loop_overhead_frac = 1 - batches x step time / the epoch's training seconds: what the permutation, the index ops, the copies
    into the replayed graph's static buffers and the epoch's host syncs cost.  Two epochs are run, the second is reported (the first
    carries hipGraph capture and first-touch allocations); no model pickle is written (model_path=None) and no recommendation test is
    run (resp_model=None): both are outside the step loop and `eval` / the checkpoint tests cover them."""
    import numpy as np
    from pivotcvae_amd.train_generative import Trainer, train_on_dataset
    args, cfg, model, device = ctx["args"], ctx["cfg"], ctx["model"], ctx["device"]
    N, S, B = cfg["N"], cfg["S"], cfg["B"]
    rng = np.random.default_rng(31)

    def dataset(n):
        return {"slates": rng.integers(0, N, size=(n, S)), "users": rng.integers(0, N_USER, size=(n, 1)),
                "responses": (rng.random((n, S)) < 0.5).astype(np.float32), "nCandidate": 1000}

    train, val = dataset(n_batches * B), dataset(val_batches * B)
    out = {"train_slates": n_batches * B, "val_slates": val_batches * B, "batch": B,
           "reference": "train_generative.py:120-165 (DataLoader(shuffle=True) -> step loop -> validation at n_neg = nCandidate)"}
    was_flag = model.candidateFlag
    try:
        for name, flag in (("candidates_1000", True), ("mask_train_n_neg_1000", False)):
            model.candidateFlag = flag
            mode = dict(n_candidate=1000) if flag else dict(n_neg=1000)
            opt = ctx["trainer"].opt   # the replica's one optimiser (a second FlatAdam would re-home the parameters)
            loop_tr = Trainer(model, lr=LR, beta=BETA, capture_graph=not args.no_graph, optimizer=opt, **mode)
            hist = train_on_dataset(train, val, model, None, _NullLogger(), None, B, 2, LR, 0.0, BETA, n_neg=1000, seed=3,
                                    trainer=loop_tr)
            del loop_tr
            tr = Trainer(model, lr=LR, beta=BETA, capture_graph=not args.no_graph, resident_batch=True, optimizer=opt, **mode)
            v = StepTimer(tr, ctx["batch"], B, ctx["lo"], False, device).run(40, 5)   # (ms-scale steps: 10 of them left the ratio at +-5 %)
            step_ms = v["dt"] / v["steps"] * 1e3
            t_train, t_val = hist["train_seconds"][-1], hist["val_seconds"][-1]
            out[name] = {"slates_per_s": n_batches * B / (t_train + t_val), "train_slates_per_s": n_batches * B / t_train,
                         "epoch_train_ms": t_train * 1e3, "epoch_val_ms": t_val * 1e3, "batches": n_batches,
                         "ms_per_batch_in_the_loop": t_train * 1e3 / n_batches, "ms_per_step_resident_batch": step_ms,
                         "loop_overhead_frac": 1.0 - n_batches * step_ms / (t_train * 1e3),
                         "val_slates_per_s": val_batches * B / t_val, "launch": "hipGraph replay" if v["graphed"] else "eager",
                         "first_epoch_train_ms": hist["train_seconds"][0] * 1e3,
                         "train_loss": hist["train"][-1], "val_loss": hist["val"][-1]}
            del tr
    finally:
        model.candidateFlag = was_flag
    return out
