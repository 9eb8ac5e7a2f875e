"""evt_graph_probe.py: can a HIP event be recorded inside a captured hipGraph?  (ROCm 7.2: hipErrorInvalidHandle - why bench.py times
the catalog kernel over eager steps after a graph-replayed timed region.)"""
import torch, time
x = torch.randn(4096, 4096, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): y = x @ x
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        e0.record()
        y = x @ x
        y = y @ x
        e1.record()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "elapsed", e0.elapsed_time(e1))
except Exception as ex:
    print("FAILED:", type(ex).__name__, str(ex)[:300])
