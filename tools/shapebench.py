"""store shapes on a slow and a fast output buffer (tools/shapebench.hip); the buffers are classified with the real C3 launch"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd"); mod.LIB_PATH = os.path.join(ROOT, "exp", "libvhp_PLAIN.so")  # (the candidates are told apart by the launch with PLAIN stores: 0.51 / 0.72 ms)
synth = import_module("visibility-heuristic-path-planner_amd.synth")
ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n, side = 256, 1000
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
c = mod.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream); c.set_map(occ); c.set_option("kernel", 3)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
sb = C.CDLL(os.path.join(ROOT, "exp", "libshapebench.so"))
sb.shape_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
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
names = ["y1k", "y1k_half", "y1k_mixed", "x128", "x64", "x256", "x512", "fill", "rand128", "rand1k", "y1k_8B", "x128_rows2", "y2k", "y1k_rot", "y1k_off1of16", "y1k_off2of16", "y1k_off4of16", "y1k_off8of16", "y1k_nohalves", "y1k_halveslater", "y1k_mis_nohalves", "y1k_lanes4to59"]
print("%-12s %4s | %9s %9s %6s" % ("pattern", "wpc", "slow TB/s", "fast TB/s", "ratio"))
todo = [(pat, nm) for pat, nm in enumerate(names) if not (len(sys.argv) > 2 and pat < int(sys.argv[2]) and pat not in (0, 2))]
if len(sys.argv) > 3:  # the same patterns with non-temporal stores
    todo = [(pat, nm) for pat, nm in enumerate(names) if nm in sys.argv[3].split(",")]
    todo = todo + [(100 + pat, nm + "+nt") for pat, nm in todo]
for pat, nm in todo:
    for wpc in (8, 12, 16):
        ms, by = C.c_float(), C.c_double()
        r = []
        for ptr in (slow, fast):
            torch.cuda.synchronize()
            assert sb.shape_run(ptr, pat, wpc, C.byref(ms), C.byref(by)) == 0
            r.append(by.value / ms.value / 1e9)
        print("%-12s %4d | %9.2f %9.2f %6.2f" % (nm, wpc, r[0], r[1], r[1] / r[0]), flush=True)
