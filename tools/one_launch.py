"""A few launches of one library build / kernel option, for rocprofv3 --pmc passes.  Diagnostic only.
usage: one_launch.py <lib or -> <kernel option> [side] [n sources] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
lib, kopt = sys.argv[1], int(sys.argv[2])
side = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
if lib != "-":
    mod.LIB_PATH = os.path.join(ROOT, lib)
lo, hi = (20, 100) if side <= 1024 else (80, 400)
occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
src = synth.free_sources(occ, n, seed=7)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", kopt)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
for _ in range(reps):
    c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
