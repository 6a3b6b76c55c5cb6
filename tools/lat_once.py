"""A few single-source launches of one kernel (for rocprofv3 --pmc runs).  usage: lat_once.py <kernel option> [side] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
k = int(sys.argv[1]); side = int(sys.argv[2]) if len(sys.argv) > 2 else 1000; n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
occ = np.ones((side, side), np.uint8)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", k)
d_src = torch.from_numpy(np.array([[side // 2, side // 2]], np.int32)).cuda()
out = torch.empty((1, side, side), dtype=torch.float64, device="cuda")
for _ in range(n):
    c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
torch.cuda.synchronize()
