#!/usr/bin/env python3
"""Instruction mix of the hottest loop of one kernel:  asm_loop_stats.py file.hip kernel_substring [--dump]"""
import re, subprocess, sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
dump = "--dump" in sys.argv
asm = "/tmp/_loopstats.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", src, "-o", asm],
               stderr=subprocess.DEVNULL, check=True)
a = open(asm).read()
starts = [m for m in re.finditer(r"^(_Z\S+):\s*; @", a, re.M)]
for m in starts:
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if pat in name.replace("(anonymous namespace)::", ""):
        break
else:
    sys.exit("kernel not found")
end = a.find(".Lfunc_end", m.end())
lines = a[m.end():end].split("\n")
mf = [n for n, l in enumerate(lines) if "v_mfma" in l]
labels = {l.strip()[:-1]: n for n, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l.strip())}
loops = []
for n, l in enumerate(lines):
    mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        lo = labels[mm.group(1)]
        loops.append((lo, n, sum(1 for x in mf if lo <= x <= n)))
for lo, hi, cnt in loops:
    print(f"  loop {lo}..{hi}: {cnt} mfma")
want = int(sys.argv[sys.argv.index("--mfma") + 1]) if "--mfma" in sys.argv else None
cands = [l for l in loops if l[2] and (want is None or l[2] == want)]
best = min(cands, key=lambda l: (-(l[2]), l[1] - l[0])) if want is None else min(cands, key=lambda l: l[1] - l[0])
print(name.replace("(anonymous namespace)::", "")[:100])
print(f"total mfma {len(mf)}; hottest loop lines {best[0]}..{best[1]} with {best[2]} mfma")
c = Counter()
body = [l.strip() for l in lines[best[0]:best[1] + 1] if l.strip() and not l.strip().startswith((";", "."))]
for t in body:
    c[t.split()[0]] += 1
tot = sum(c.values())
print(f"{tot} instructions:", ", ".join(f"{op} {n}" for op, n in c.most_common(28)))
if dump:
    print("\n".join(body))
