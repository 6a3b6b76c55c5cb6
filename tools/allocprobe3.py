"""Sweep time against the offset of the output inside one big allocation.  Diagnostic only."""
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
big = torch.empty(n * 8_000_000 + (1 << 30) + 4096, dtype=torch.uint8, device="cuda")
base = big.data_ptr()
print("base %#x" % base)
def run(ptr):
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, ptr, dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(30):
        ctx.sweep_batch_device(d_src.data_ptr(), n, ptr, dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(30)
    ctx.timing(False)
    return float(np.median(k))
offs = [0, 128, 4096, 1 << 16, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20, 1 << 30,
        8_000_000, 3 * 8_000_000, 0]
for o in offs:
    print("offset %11d (%#11x): %.4f ms" % (o, o, run(base + o)), flush=True)
