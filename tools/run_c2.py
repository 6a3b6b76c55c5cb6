"""Tiny driver for profiling: N launches of the sweep for a chosen workload (no timing logic)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if wl == "c2":
    occ = np.ones((1000, 1000), np.uint8); src = np.array([[500, 500]], np.int32)
else:
    occ, src = synth.config_c3(256)
ctx = vhp_amd.Context(0); ctx.set_map(occ)
d_src = torch.from_numpy(src).cuda()
d_out = torch.empty((len(src), 1000, 1000), dtype=torch.float64, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(n):
    ctx.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
ctx.sync()
