"""Sweep time on six successive 2 GB output allocations of one process (all held).  Diagnostic only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = 256
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
def run(out):
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(30):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(30)
    ctx.timing(False)
    return float(np.median(k))
bufs = []
for i in range(6):
    bufs.append(torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda"))
for rep in range(2):
    print(" ".join("%d:%.3f" % (i, run(b)) for i, b in enumerate(bufs)), flush=True)
def fill_rate(b):
    for _ in range(2): b.fill_(1.0)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): b.fill_(1.0)
    e.record(); torch.cuda.synchronize()
    return b.numel() * 8 * 10 / a.elapsed_time(e) / 1e9
print("fill_ TB/s:", " ".join("%d:%.2f" % (i, fill_rate(b)) for i, b in enumerate(bufs)))
print("ptrs", " ".join("%#x" % b.data_ptr() for b in bufs))
