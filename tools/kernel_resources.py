#!/usr/bin/env python3
"""Compact per-kernel register/LDS/occupancy table for one .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'vgpr':>5}{'agpr':>5}{'sgpr':>5}{'spill':>6}{'scr':>5}{'occ':>4}{'lds':>7}  kernel")
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*\)$", "", n)
    if filt and filt not in n:
        continue
    print(f"{r.get('vgpr',0):>5}{r.get('agpr',0):>5}{r.get('sgpr',0):>5}{r.get('spill',0):>6}{r.get('scratch',0):>5}{r.get('occ',0):>4}{r.get('lds',0):>7}  {n}")
