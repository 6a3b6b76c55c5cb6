"""A/B of library builds over SOURCE POSITIONS: one-source launches of the latency sweep from k seeded free cells of the
random-rectangle map (or of an empty grid: AB_EMPTY=1), every build timed on every source in alternation.  One source says little
about a change to the latency sweep -- which octant is the launch's longest depends on where the source sits.  Diagnostic only.
usage: ab_positions.py <side> <k sources> <lib> ...      ("-" = the in-tree build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
side, k = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
if os.environ.get("AB_EMPTY"):
    occ = np.ones((side, side), np.uint8)
else:
    lo, hi = (20, 100) if side <= 1024 else (80, 400)
    occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
src = synth.free_sources(occ, k, seed=11)
out = torch.empty((1, side, side), dtype=torch.float64, device="cuda")
ctxs = []
for lib in libs:
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if lib == "-" else os.path.join(ROOT, lib)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    c.set_option("kernel", 4)
    ctxs.append(c)
tab = np.zeros((k, len(libs)))
for i in range(k):
    d_src = torch.from_numpy(np.ascontiguousarray(src[i:i + 1], np.int32)).cuda()
    acc = [[] for _ in libs]
    for rep in range(4):
        for j, ctx in enumerate(ctxs):
            for _ in range(3):
                ctx.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
            torch.cuda.synchronize()
            ctx.timing(True)
            for _ in range(15):
                ctx.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
            torch.cuda.synchronize()
            t = ctx.timing_collect(15)
            ctx.timing(False)
            if rep:
                acc[j].append(float(np.median(t)))
    tab[i] = [np.median(a) * 1e3 for a in acc]
    print("source (%4d, %4d): " % tuple(src[i]) + "  ".join("%7.1f" % v for v in tab[i]))
print("side %d, %d sources, us per launch -- mean: " % (side, k) + "  ".join("%s %.1f" % (l, v) for l, v in zip(libs, tab.mean(0))))
