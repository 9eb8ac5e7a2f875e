#!/usr/bin/env python3
"""Time the fused catalog CE call alone (config-4 shape by default), optionally for several builds of the library:

    python tools/bench_catalog.py [--R 81920 --N 1000000 --D 128 --dtype bf16 --iters 10] [--libs a.so b.so ...]

With --libs each library is timed in its own subprocess (PCVAE_LIB), interleaved over --rounds rounds.
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    import torch
    from pivotcvae_amd import ops
    from pivotcvae_amd._hip import PREC_NAMES
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    E = torch.rand(args.N, args.D, device=dev, generator=g) * 2 - 1
    E = E / E.norm(dim=1, keepdim=True)
    rx = (torch.rand(args.R, args.D, device=dev, generator=g) * 2 - 1) * args.scale
    tgt = torch.randint(0, args.N, (args.R,), device=dev, generator=g)
    table = ops.CatalogTable(E)
    prec = PREC_NAMES[args.dtype]
    for _ in range(2):
        nll, lse, dx = ops.catalog_ce_raw(rx, table, tgt, prec=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        nll, lse, dx = ops.catalog_ce_raw(rx, table, tgt, prec=prec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    tf = 4.0 * args.R * args.N * args.D / (ms * 1e-3) / 1e12
    print(json.dumps({"ms": ms, "tflops": tf, "nll_mean": float(nll.mean()), "dx_abs_mean": float(dx.abs().mean())}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=81920)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--libs", nargs="*")
    a = ap.parse_args()
    if not a.libs:
        one(a)
    else:
        res = {l: [] for l in a.libs}
        base = [sys.executable, os.path.abspath(__file__), "--R", str(a.R), "--N", str(a.N), "--D", str(a.D), "--dtype", a.dtype,
                "--iters", str(a.iters), "--scale", str(a.scale)]
        for _ in range(a.rounds):
            for l in a.libs:
                out = subprocess.run(base, env=dict(os.environ, PCVAE_LIB=os.path.abspath(l)), capture_output=True, text=True)
                line = [x for x in out.stdout.splitlines() if x.startswith("{")]
                res[l].append(json.loads(line[-1]) if line else {"error": out.stderr[-300:]})
        for l, v in res.items():
            print(l, " | ".join(f"{x.get('ms', -1):.2f} ms {x.get('tflops', 0):.0f} TF nll {x.get('nll_mean', 0):.5f}" if "ms" in x else str(x) for x in v))
