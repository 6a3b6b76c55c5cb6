"""Which stores suffer on a 'slow' output allocation?  Six allocations x {all stores, y-major only, x-major only,
none} (diagnostic builds exp/libvhp_NOSTORE_X.so etc.), one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = 256
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
libs = ["-", "exp/libvhp_NOSTORE_X.so", "exp/libvhp_NOSTORE_Y.so", "exp/libvhp_NOSTORE.so"]
ctxs = []
for lib in libs:
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if lib == "-" else os.path.join(ROOT, lib)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    ctxs.append(c)
def run(ctx, out):
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(25):
        ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(25)
    ctx.timing(False)
    return float(np.median(k))
bufs = [torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda") for _ in range(6)]
for lib, ctx in zip(libs, ctxs):
    print("%-28s" % lib, " ".join("%d:%.3f" % (i, run(ctx, b)) for i, b in enumerate(bufs)), flush=True)
