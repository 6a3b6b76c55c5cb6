"""Library builds on a SLOW and on a FAST output buffer of one process: N hipMalloc'ed candidates are
classified with the first library; every library is then timed on the slowest and on the fastest.  Diagnostic only.
usage: ab_slowfast.py <n candidates> <n sources> <lib>[@key=value,...] ...      ("-" = the in-tree build)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
ncand, n = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
side = 1000
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
ctxs = []
for lib in libs:
    path, _, shape = lib.partition("@")
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if path == "-" else os.path.join(ROOT, path)
    c = mod.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_map(occ)
    c.set_option("kernel", 3)
    for kv in (shape.split(",") if shape else []):
        k, v = kv.split("=")
        c.set_option(k, int(v))
    ctxs.append(c)
def run(c, ptr, reps):
    for _ in range(2):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    c.timing(True)
    for _ in range(reps):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    t = c.timing_collect(reps)
    c.timing(False)
    return float(np.median(t))
bufs = []
for i in range(ncand):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n * side * side * 8) == 0
    bufs.append(p.value)
t0 = [run(ctxs[0], p, 5) for p in bufs]
print("candidates (first library), launch ms:", " ".join("%.3f" % t for t in t0), flush=True)
slow, fast = bufs[int(np.argmax(t0))], bufs[int(np.argmin(t0))]
print("%-60s %9s %9s %7s" % ("library", "slow ms", "fast ms", "ratio"))
for lib, c in zip(libs, ctxs):
    a = [run(c, slow, 12) for _ in range(2)]
    b = [run(c, fast, 12) for _ in range(2)]
    print("%-60s %9.3f %9.3f %7.2f" % (lib, min(a), min(b), min(a) / min(b)), flush=True)
# coverage: which share of the cells does the LAST library write at all?  (diagnostic builds that drop stores)
if os.environ.get("AB_COVERAGE"):
    t = torch.full((n, side, side), float("nan"), dtype=torch.float64, device="cuda")
    for lib, c in zip(libs, ctxs):
        t.fill_(float("nan"))
        c.sweep_batch_device(d_src.data_ptr(), n, t.data_ptr())
        torch.cuda.synchronize()
        print("%-60s writes %.2f %% of the cells" % (lib, 100.0 * float((~torch.isnan(t)).sum()) / t.numel()), flush=True)
