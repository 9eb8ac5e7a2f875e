#!/usr/bin/env python3
"""Time the candidate-set reconstruction term alone (config-4 shape by default): the fused kernel (pcvae_candidate_ce, sets drawn
in-kernel and sets given), the materialised route it replaces (candidate_draw -> [R, Cn] ids -> K9 scores -> dense CE -> K9
backward) and the sparse mask-train kernel over the same gather volume (n_neg = Cn), optionally for several builds of the library:

    python tools/bench_candidate.py [--R 81920 --N 1000000 --D 128 --Cn 1000 --iters 10] [--libs a.so b.so ...]
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters):
    import torch
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def one(args):
    import torch
    from pivotcvae_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    E = torch.rand(args.N, args.D, device=dev, generator=g) * 2 - 1
    E = E / E.norm(dim=1, keepdim=True)
    rx = (torch.rand(args.R, args.D, device=dev, generator=g) * 2 - 1) * args.scale
    feat = torch.randint(0, args.N, (args.R,), device=dev, generator=g)
    table = ops.CatalogTable(E)
    R, Cn, D = args.R, args.Cn, args.D
    res = {"R": R, "N": args.N, "D": D, "Cn": Cn, "requested_GB": R * Cn * D * 4 / 1e9}
    from pivotcvae_amd._hip import PREC_BF16, PREC_F32
    prec = PREC_BF16 if args.bf16 else PREC_F32
    res["rows"] = "bf16" if args.bf16 else "f32"
    if args.bf16:
        res["requested_GB"] /= 2
    ms, (nll, lse, dx, _) = timed(lambda: ops.candidate_ce_raw(rx, table, Cn, feat, 7, 0, prec=prec), args.iters)
    if args.bf16:
        res["fused_drawn_ms"] = ms
        res["fused_drawn_TBps_requested"] = R * Cn * D * 2 / (ms * 1e-3) / 1e12
        ms2, _ = timed(lambda: ops.catalog_ce_sparse_raw(rx, table, feat, Cn / args.N, seed=7, prec=prec), args.iters)
        res["sparse_n_neg_ms"] = ms2
        print(json.dumps(res))
        return
    res["fused_drawn_ms"] = ms
    res["fused_drawn_TBps_requested"] = R * Cn * D * 4 / (ms * 1e-3) / 1e12
    res["nll_mean"] = float(nll.mean())
    ms_fwd, _ = timed(lambda: ops.candidate_ce_raw(rx, table, Cn, feat, 7, 0, want_dx=False), args.iters)
    res["fused_drawn_fwd_only_ms"] = ms_fwd
    if not args.fused_only:
        cand, tgt = ops.candidate_draw(feat.view(R, 1), args.N, Cn, seed=7, row_offset=0)
        cand, tgt = cand.view(R, Cn), tgt.view(R)
        ms, (n2, _, d2, _) = timed(lambda: ops.candidate_ce_raw(rx, table, cand=cand, cand_target=tgt), args.iters)
        res["fused_given_ms"] = ms
        res["given_equals_drawn"] = bool(torch.equal(n2, nll) and torch.equal(d2, dx))

        def materialised():
            c, t = ops.candidate_draw(feat.view(R, 1), args.N, Cn, seed=7, row_offset=0)
            rd = rx.detach().requires_grad_(True)
            loss = ops.dense_ce(ops.candidate_scores(rd, E, c.view(R, Cn)), t.view(R))
            loss.backward()
            return loss, rd.grad
        ms, (lm, gm) = timed(materialised, max(2, args.iters // 3))
        res["materialised_ms"] = ms
        res["materialised_loss"] = float(lm)
        res["materialised_vs_fused_dx_max_abs"] = float((gm * R - dx).abs().max())
        ms, _ = timed(lambda: ops.catalog_ce_sparse_raw(rx, table, feat, Cn / args.N, seed=7), args.iters)
        res["sparse_n_neg_ms"] = ms
    print(json.dumps(res))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=81920)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--Cn", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--fused-only", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="gather rows of the bf16 table (configs 3 / 5's stated arithmetic)")
    ap.add_argument("--libs", nargs="*")
    a = ap.parse_args()
    if not a.libs:
        one(a)
    else:
        base = [sys.executable, os.path.abspath(__file__), "--R", str(a.R), "--N", str(a.N), "--D", str(a.D), "--Cn", str(a.Cn),
                "--iters", str(a.iters), "--scale", str(a.scale), "--fused-only"] + (["--bf16"] if a.bf16 else [])
        for _ in range(a.rounds):
            for l in a.libs:
                out = subprocess.run(base, env=dict(os.environ, PCVAE_LIB=os.path.abspath(l)), capture_output=True, text=True)
                line = [x for x in out.stdout.splitlines() if x.startswith("{")]
                print(l, line[-1] if line else out.stderr[-300:], flush=True)
