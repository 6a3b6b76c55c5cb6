"""Is the fast/slow state of the sweep a property of the output buffer?  Times buffer A, then B (allocated while
A is held), then A again, B again.  Diagnostic only."""
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
def run(out, tag):
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(40):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(40)
    ctx.timing(False)
    print("%s ptr %#x: median %.4f ms (first 5: %s)" % (tag, out.data_ptr(), np.median(k), " ".join("%.3f" % x for x in k[:5])), flush=True)
A = torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda")
run(A, "A (first allocation)")
run(A, "A again")
B = torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda")
run(B, "B (second allocation)")
run(A, "A again")
run(B, "B again")
import time
time.sleep(2.0)
run(A, "A after 2 s idle")
run(B, "B after idle")
A.zero_(); torch.cuda.synchronize()
run(A, "A after zero_")
