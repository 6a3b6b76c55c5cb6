"""A/B of sweep launch shapes inside one process (same box, same clocks): alternates the VHP_R / VHP_W /
VHP_MULTI overrides between blocks of launches and prints the median kernel time per shape.  Diagnostic only.
usage: ab_shapes.py <workload side> <n sources> "R W [MULTI]" "R W [MULTI]" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
side, n = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:]
lo, hi = (20, 100) if side <= 1024 else (80, 400)
occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
res = {c: [] for c in cfgs}
for rep in range(6):
    for c in cfgs:
        parts = c.split()
        for k in ("VHP_R", "VHP_W", "VHP_MULTI", "VHP_SLIDE"):
            os.environ.pop(k, None)
        if parts[0].startswith("slide"):
            os.environ["VHP_SLIDE"] = parts[0][5:]
        elif parts[0] != "-":
            os.environ["VHP_R"], os.environ["VHP_W"] = parts[0], parts[1]
            if len(parts) > 2:
                os.environ["VHP_MULTI"] = parts[2]
        for _ in range(3):
            ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
        torch.cuda.synchronize()
        ctx.timing(True)
        for _ in range(25):
            ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
        torch.cuda.synchronize()
        k = ctx.timing_collect(25)
        ctx.timing(False)
        if rep:  # the first block of a process runs ~15% faster than steady state on this pool
            res[c].append(float(np.median(k)))
for c in cfgs:
    v = res[c]
    print("side %d n %d shape [%s]: median %.4f ms (blocks: %s)  %.0f GB/s" % (side, n, c, np.median(v), " ".join("%.3f" % x for x in v),
          9.0 * side * side * n / np.median(v) / 1e6))
