"""for rocprofv3 --pmc: the C3 launch on candidates (classification), then 6 launches on the slowest and 6 on the fastest buffer:
the LAST 12 dispatches of vhp_pool_sweep in the counter file are slow x 6, fast x 6.  Diagnostic only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, side = 256, 1000
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
c = mod.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream); c.set_map(occ); c.set_option("kernel", 3)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
def run(ptr, reps):
    c.timing(True)
    for _ in range(reps): c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize(); t = c.timing_collect(reps); c.timing(False)
    return float(np.median(t))
bufs = []
for i in range(ncand):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n * side * side * 8) == 0; bufs.append(p.value)
t0 = [run(p, 3) for p in bufs]
slow, fast = bufs[int(np.argmax(t0))], bufs[int(np.argmin(t0))]
print("candidates:", " ".join("%.3f" % t for t in t0))
print("slow x 6: %.3f ms   fast x 6: %.3f ms (under the profiler)" % (run(slow, 6), run(fast, 6)))
