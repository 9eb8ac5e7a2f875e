#!/usr/bin/env python3
"""Time the catalog argmax (K6) alone: exact f32-MFMA kernel vs bf16-screened exact route, and check they agree.

    python tools/bench_argmax.py [--R 8192 --N 1000000 --iters 5]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=8192)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    import torch
    from pivotcvae_amd import ops
    dev, D = "cuda:0", 128
    g = torch.Generator(device=dev).manual_seed(0)
    E = torch.rand(a.N, D, device=dev, generator=g) * 2 - 1
    E = E / E.norm(dim=1, keepdim=True)
    x = (torch.rand(a.R, D, device=dev, generator=g) * 2 - 1) * a.scale
    table = ops.CatalogTable(E)
    out = {}
    ids = {}
    for name, scr in (("f32", False), ("screened", True)):
        for _ in range(2):
            ids[name] = ops.catalog_argmax(x, table, return_best=True, screened=scr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.catalog_argmax(x, table, screened=scr)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        out[name] = {"ms": ms, "tflops_algorithmic": 2.0 * a.R * a.N * D / (ms * 1e-3) / 1e12}
    out["ids_equal"] = bool(torch.equal(ids["f32"][0], ids["screened"][0]))
    out["best_equal"] = bool(torch.equal(ids["f32"][1], ids["screened"][1]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
