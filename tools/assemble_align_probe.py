#!/usr/bin/env python3
"""assemble_inputs_vec_kernel (the train step's own gather: item / user / pivot rows + one-hot click count into the three stacks'
inputs, ONE launch) at config 4 with PACKED input rows (ld = 1419 / 139 / 283 floats: rows start on arbitrary 4-byte boundaries) and
with rows padded to cache-line multiples (PCVAE_ASSEMBLE_ROW_ALIGN floats), cold caches, the dispatch's own HIP events.

    python tools/assemble_align_probe.py > profiles/r06_assemble_row_align_probe.txt

Bytes per launch (read once + written, indices included) are the same in every variant: what changes is how many of the written
cache lines are PARTIAL.  On the SAME box, the same way (cold caches, the dispatch's own events where the library has them, a HIP
event pair around ONE launch otherwise): the standalone gather kernel on the same rows (gather_rows_coal_kernel: aligned, contiguous
output) and a plain device-to-device copy of the same number of bytes - the box's own streaming rate, the ceiling of any kernel
that reads N bytes and writes N bytes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
sys.modules.setdefault("bench", bench)
import bench_extras     # noqa: E402
from pivotcvae_amd import ops   # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "4"]
dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, dev, "f32")
s, r, u = bench.synthetic_batch(cfg, cfg["B"], dev)
B, S, D = cfg["B"], cfg["S"], cfg["D"]
flush = torch.empty(128 * 1024 * 1024, device=dev)
print(f"# config N={cfg['N']} S={S} D={D} B={B}; 13 cold launches per variant (512 MB written before each), first 3 dropped")
for align in (1, 4, 16, 32, 64):
    ops.ASSEMBLE_ROW_ALIGN = align
    ts, nbytes = [], None

    def hook_end(tok, nb):
        global nbytes
        nbytes = nb

    for it in range(13):
        flush.fill_(float(it))
        torch.cuda.synchronize()
        ops.ASSEMBLE_TIMING = (lambda: None, hook_end)
        d = bench_extras.kernel_timer_run(lambda: ops.assemble_inputs(model.docEmbed.weight, model.userEmbed.weight, s, r, u, bench.Z),
                                          bench_extras.TIMER_ASSEMBLE)
        ops.ASSEMBLE_TIMING = None
        if it >= 3:
            ts += d
    ts.sort()
    med, mean = ts[len(ts) // 2], sum(ts) / len(ts)
    enc_ld = ops.assemble_inputs(model.docEmbed.weight, model.userEmbed.weight, s, r, u, bench.Z)[0].stride(0)
    print(f"row align {align:3d} floats (enc_in ld {enc_ld}): median {med * 1e3:6.2f} us  mean {mean * 1e3:6.2f} us  min {ts[0] * 1e3:6.2f}  "
          f"max {ts[-1] * 1e3:6.2f}   {nbytes / 1e6:.1f} MB -> {nbytes / (med * 1e-3) / 1e12:.3f} TB/s = {nbytes / (med * 1e-3) / 8e12:.3f} of 8 TB/s")

# ---- the same box's reference points
ops.ASSEMBLE_ROW_ALIGN = 1
n_idx = B * (S + 2)
idx = torch.randint(0, cfg["N"], (n_idx,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
out = torch.empty(n_idx, D, device=dev)
ts = []
for it in range(13):
    flush.fill_(float(it))
    torch.cuda.synchronize()
    d = bench_extras.kernel_timer_run(lambda: ops.gather_rows(model.docEmbed.weight, idx, out=out), bench_extras.TIMER_GATHER)
    if it >= 3:
        ts += d
ts.sort()
gb = n_idx * (2 * D * 4 + 8)
print(f"gather_rows_coal_kernel, same rows ({n_idx} x {D * 4} B, contiguous aligned output): median {ts[len(ts) // 2] * 1e3:6.2f} us   "
      f"{gb / 1e6:.1f} MB -> {gb / (ts[len(ts) // 2] * 1e-3) / 8e12:.3f} of 8 TB/s")
half = 110_900_000 // 8 * 4
src, dst = torch.empty(half // 4, device=dev), torch.empty(half // 4, device=dev)
ts = []
for it in range(13):
    flush.fill_(float(it))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    if it >= 3:
        ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"plain copy of {half / 1e6:.1f} MB (read) + {half / 1e6:.1f} MB (written), event pair around one launch (carries the pair's ~2.4 us): "
      f"median {ts[len(ts) // 2] * 1e3:6.2f} us -> {2 * half / (ts[len(ts) // 2] * 1e-3) / 8e12:.3f} of 8 TB/s")
