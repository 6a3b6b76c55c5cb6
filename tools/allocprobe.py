"""Does the allocator matter?  Six 2 GB output buffers from hipMalloc (ctypes) and six from torch, timed in turn."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = 256
occ = synth.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
ctx.set_option("kernel", int(os.environ.get("VHP_KERNEL", "0")))
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
def run(ptr):
    for _ in range(3):
        ctx.sweep_batch_device(d_src.data_ptr(), n, ptr, dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(30):
        ctx.sweep_batch_device(d_src.data_ptr(), n, ptr, dtype=vhp_amd.F64)
    torch.cuda.synchronize()
    k = ctx.timing_collect(30)
    ctx.timing(False)
    return float(np.median(k))
raw = []
for i in range(6):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n * 8_000_000) == 0
    raw.append(p.value)
tb = [torch.empty((n, 1000, 1000), dtype=torch.float64, device="cuda") for _ in range(6)]
for rep in range(2):
    print("hipMalloc:", " ".join("%d:%.3f" % (i, run(p)) for i, p in enumerate(raw)), flush=True)
    print("torch    :", " ".join("%d:%.3f" % (i, run(t.data_ptr())) for i, t in enumerate(tb)), flush=True)
