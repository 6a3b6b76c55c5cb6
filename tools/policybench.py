"""cache-policy bits of the stores against the cost of partially written lines, on a slow and a fast buffer (tools/policybench.hip)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, side = 256, 1000
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
mod.LIB_PATH = os.path.join(ROOT, "exp", "libvhp_PLAIN.so")  # (the candidates are told apart by the launch with PLAIN stores: 0.51 / 0.72 ms)
c = mod.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream); c.set_map(occ); c.set_option("kernel", 3)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
pb = C.CDLL(os.path.join(ROOT, "exp", "libpolicybench.so"))
pb.policy_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
def run(ptr, reps=5):
    for _ in range(2): c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize(); c.timing(True)
    for _ in range(reps): c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize(); t = c.timing_collect(reps); c.timing(False)
    return float(np.median(t))
bufs = []
for i in range(ncand):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n * side * side * 8 + 65536) == 0; bufs.append(p.value)
t0 = [run(p) for p in bufs]
print("C3 launch ms per candidate:", " ".join("%.3f" % t for t in t0), flush=True)
slow, fast = bufs[int(np.argmax(t0))], bufs[int(np.argmin(t0))]
names = ["plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
print("%-12s %-22s | %9s %9s" % ("policy", "pattern", "slow TB/s", "fast TB/s"))
for pol, nm in enumerate(names):
    for aligned, halves, what in ((1, 0, "all on lines"), (0, 0, "odd rows off, all lanes"), (0, 1, "odd rows off, halves only")):
        if pol == 0 and halves: continue
        ms, by = C.c_float(), C.c_double()
        r = []
        for ptr in (slow, fast):
            torch.cuda.synchronize()
            assert pb.policy_run(ptr, pol, aligned, halves, 12, C.byref(ms), C.byref(by)) == 0
            r.append(by.value / ms.value / 1e9)
        print("%-12s %-26s | %9.2f %9.2f" % (nm, what, r[0], r[1]), flush=True)
