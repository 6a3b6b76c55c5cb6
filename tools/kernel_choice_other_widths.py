"""The three batch kernels on one launch each (latency / pool / front sweep, kernel time in us) and the library's own choice: for re-measuring the kernel-choice rule
(csrc/vhp_capi.hip use_lat_kernel / use_pool_kernel).  Diagnostic only.  usage: kernel_choice_other_widths.py NXxNYxSOURCES ..."""
import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
for (nx, ny, n) in [(int(a), int(b), int(c)) for a, b, c in (t.split("x") for t in sys.argv[1:])]:
    lo, hi = (3, max(nx // 8, 4))
    occ = synth.random_rect_map(nx, ny, 30, lo, hi, lo, max(ny // 8, 4), seed=1)
    src = synth.free_sources(occ, n, seed=7)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    out = torch.empty((n, ny, nx), dtype=torch.float64, device="cuda")
    res = []
    for k in (4, 3, 1):
        c = mod.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream); c.set_map(occ); c.set_option("kernel", k)
        try:
            for _ in range(3): c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize(); c.timing(True)
            for _ in range(30): c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
            torch.cuda.synchronize()
            res.append("%d: %.1f" % (k, 1e3 * np.median(c.timing_collect(30))))
        except Exception as e:
            res.append("%d: refused" % k)
    c = mod.Context(0); c.set_map(occ); c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr()); torch.cuda.synchronize()
    print("%4d x %4d, %3d sources: %s   (library's choice: %d)" % (nx, ny, n, "  ".join(res), c.last_sweep_kernel()))
