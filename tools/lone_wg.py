"""Latency of a lone large quadrant: one corner source on the C3 map, streaming kernel (and the front sweep for
comparison).  Diagnostic only.  usage: lone_wg.py <lib or -> ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
occ[5, 5] = 1
for lib in sys.argv[1:]:
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if lib == "-" else os.path.join(ROOT, lib)
    for kopt in (2, 22, 1):  # streaming sweep, the same with two tile slots, front sweep
        for srcs in ([[5, 5]], [[500, 500]], [[5, 5]] * 8):
            src = np.array(srcs, np.int32)
            n = len(src)
            c = mod.Context(0)
            c.set_stream(torch.cuda.current_stream().cuda_stream)
            c.set_map(occ)
            c.set_option("kernel", 2 if kopt == 22 else kopt)
            if kopt == 22:
                c.set_option("stream_tile_slots", 2)
            d_src = torch.from_numpy(src).cuda()
            out = torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda")
            for _ in range(3):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize()
            c.timing(True, prealloc=24)
            for _ in range(20):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize()
            k = c.timing_collect(20)
            print("lib %-28s kernel %d sources %-22s: median %.1f us" % (lib, kopt, str(srcs[0]) + "x%d" % n, 1e3 * float(np.median(k))), flush=True)
