"""The latency sweep against the front sweep for small batches: kernel time by grid side and source count.  Diagnostic only.
usage: lat_vs_front.py [side ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
sides = [int(a) for a in sys.argv[1:]] or [256, 512, 1000, 2048]
for side in sides:
    lo, hi = (max(side // 50, 2), max(side // 10, 4))
    occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    for n in (1, 2, 4, 8, 16, 32, 64):
        if n * side * side * 8 > 8e9:
            continue
        src = synth.free_sources(occ, n, seed=7)
        if n == 1:
            src = np.array([[side // 2, side // 2]], np.int32)
            occ1 = occ.copy(); occ1[side // 2, side // 2] = 1; c.set_map(occ1)
        else:
            c.set_map(occ)
        d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
        out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
        res = {}
        for k in (1, 4):
            c.set_option("kernel", k)
            for _ in range(3):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize()
            c.timing(True)
            for _ in range(15):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize()
            res[k] = float(np.median(c.timing_collect(64)))
            c.timing(False)
        print("side %5d  sources %3d: front %8.1f us  latency sweep %8.1f us  (%.2f)" % (side, n, 1e3 * res[1], 1e3 * res[4], res[4] / res[1]))
