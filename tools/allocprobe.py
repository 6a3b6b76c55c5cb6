"""Does the sweep's kernel time depend on where the output buffer lands?  Diagnostic only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
keep = []
for trial in range(8):
    if trial == 4:
        keep.clear(); torch.cuda.empty_cache()
    pad = torch.empty(((trial % 4) * 3 + 1) * 1000 * 1000, dtype=torch.float64, device="cuda") if trial % 2 else None
    out = torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda")
    keep.append((pad, out))
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(30):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(30)
    ctx.timing(False)
    print("trial %d ptr %#x (mod 2MiB %#x)  kernel ms: mean %.4f min %.4f max %.4f" % (trial, out.data_ptr(), out.data_ptr() % (2 << 20), k.mean(), k.min(), k.max()), flush=True)
