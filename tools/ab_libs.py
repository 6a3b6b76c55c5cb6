"""A/B of library builds inside one process (same box, same clocks): loads several libvhp_hip.so builds side by
side, alternates blocks of launches between them and prints the median kernel time per build.  Diagnostic only.
usage: ab_libs.py <side> <n sources> <lib>[@key=value,...] ...      ("-" = the in-tree build; the keys are vhp_set_option
keys, e.g. -@kernel=1 exp/libvhp_NOSTORE.so@kernel=2, or the launch shapes: -@rows_per_lane=1,strips=8).
Set AB_DTYPE=f32 in the environment for fp32 fields."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
side, n = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
if side == 0:  # the C2 case: empty 1000^2, centre source
    side = 1000
    occ = np.ones((side, side), np.uint8)
    src = np.array([[500, 500]] * n, np.int32)
else:
    lo, hi = (20, 100) if side <= 1024 else (80, 400)
    occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
    src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
F32 = os.environ.get("AB_DTYPE", "f64") == "f32"
out = torch.empty((n, side, side), dtype=torch.float32 if F32 else torch.float64, device="cuda")
ctxs = []
shapes = {}
for lib in libs:
    path, _, shape = lib.partition("@")
    shapes[lib] = [kv.split("=") for kv in shape.split(",")] if shape else []
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if path == "-" else os.path.join(ROOT, path)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    for k, v in shapes[lib]:
        c.set_option(k, int(v))
    ctxs.append(c)
res = {l: [] for l in libs}
for rep in range(6):
    for lib, ctx in zip(libs, ctxs):
        for _ in range(3):
            ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F32 if F32 else mod.F64)
        torch.cuda.synchronize()
        ctx.timing(True)
        for _ in range(25):
            ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F32 if F32 else mod.F64)
        torch.cuda.synchronize()
        k = ctx.timing_collect(25)
        ctx.timing(False)
        if rep:
            res[lib].append(float(np.median(k)))
for lib in libs:
    v = res[lib]
    print("side %d n %d lib [%s]: median %.4f ms (blocks: %s)  %.0f GB/s" % (side, n, lib, np.median(v), " ".join("%.3f" % x for x in v),
          (5.0 if F32 else 9.0) * side * side * n / np.median(v) / 1e6))
