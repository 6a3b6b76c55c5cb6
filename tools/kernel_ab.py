"""Two (or more) of the batch kernels on the same map, sources and output buffer: median HIP-event time of a launch.  Diagnostic only.
usage: kernel_ab.py KERNELS SOURCE_COUNTS NXxNY [NXxNY ...] [f32]      e.g.  kernel_ab.py 1,3 64,128,256 1002x1000 1001x971
A kernel may carry launch options: 3@pool_contexts=8@pool_heads=2 (vhp_set_option keys; reset to 0 for the next kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
NAMES = {0: "auto", 1: "front", 3: "pool", 4: "latency"}
specs = sys.argv[1].split(",")
kernels = list(range(len(specs)))
KERNEL_OF = {i: int(sp.split("@")[0]) for i, sp in enumerate(specs)}
OPTS_OF = {i: [kv.split("=") for kv in sp.split("@")[1:]] for i, sp in enumerate(specs)}
NAMES = {i: NAMES[KERNEL_OF[i]] + "".join("@%s=%s" % (k, v) for k, v in OPTS_OF[i]) for i in kernels}
counts = [int(k) for k in sys.argv[2].split(",")]
f32 = "f32" in sys.argv[3:]
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[3:] if a != "f32"]
for nx, ny in sizes:
    side = max(nx, ny)
    lo, hi = (max(side // 50, 2), max(side // 10, 4))
    occ = synth.random_rect_map(nx, ny, 50, lo, min(hi, nx // 2), lo, min(hi, ny // 2), seed=1)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    for n in counts:
        if n * nx * ny * 8 > 40e9:
            continue
        src = synth.free_sources(occ, n, seed=7)
        d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
        out = torch.empty((n, ny, nx), dtype=torch.float32 if f32 else torch.float64, device="cuda")
        res, ran, ref = {}, {}, None
        for k in kernels:
            c.set_option("kernel", KERNEL_OF[k])
            for i2 in kernels:
                for kk, _ in OPTS_OF[i2]:
                    c.set_option(kk, 0)
            for kk, vv in OPTS_OF[k]:
                c.set_option(kk, int(vv))
            for _ in range(3):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F32 if f32 else mod.F64)
            torch.cuda.synchronize()
            ran[k] = c.last_sweep_kernel()
            if ref is None:
                ref = out.clone()
            elif not torch.equal(ref, out):
                print("  !! kernel %d differs from kernel %d" % (k, kernels[0]))
            c.timing(True)
            for _ in range(15):
                c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=mod.F32 if f32 else mod.F64)
            torch.cuda.synchronize()
            res[k] = float(np.median(c.timing_collect(64)))
            c.timing(False)
        print("%5d x %5d %s  sources %4d: " % (nx, ny, "f32" if f32 else "f64", n) +
              "   ".join("%s%s %8.1f us" % (NAMES[k], "" if ran[k] == KERNEL_OF[k] or KERNEL_OF[k] == 0 else "(ran %d)" % ran[k], 1e3 * res[k]) for k in kernels) +
              "   (%s / %s = %.2f)" % (NAMES[kernels[-1]], NAMES[kernels[0]], res[kernels[-1]] / res[kernels[0]]))
        del out
