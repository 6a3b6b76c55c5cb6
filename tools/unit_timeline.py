"""When every unit of a pool-sweep launch was installed and when its last strip finished (-DVHP_DIAG_TIMELINE build): which units the
launch ends on, how long a unit of a given march takes, how many generations of units a context runs.  Diagnostic only.
usage: unit_timeline.py <lib> [n sources] [n candidate buffers] [side] [key=value ...]      (vhp_set_option keys)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
lib = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
side = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
opts = [a.split('=') for a in sys.argv[5:]]
mod.LIB_PATH = os.path.join(ROOT, lib)
lo_, hi_ = (20, 100) if side >= 1000 else (max(side // 50, 2), max(side // 10, 4))
occ = synth.random_rect_map(side, side, 50, lo_, hi_, lo_, hi_, seed=1)
src = synth.free_sources(occ, n, seed=7)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", 3)
for k_, v_ in opts:
    c.set_option(k_, int(v_))
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
bufs = []
for i in range(nbuf):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n * side * side * 8) == 0
    bufs.append(p.value)
def run(ptr, reps):
    for _ in range(2):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    c.timing(True)
    for _ in range(reps):
        c.sweep_batch_device(d_src.data_ptr(), n, ptr)
    torch.cuda.synchronize()
    k = c.timing_collect(reps)
    c.timing(False)
    return float(np.median(k))
times = [run(p, 5) for p in bufs]
order = np.argsort(times)
for name, k in (("slowest", int(order[-1])), ("fastest", int(order[0]))):
    ms = run(bufs[k], 1)
    u = np.zeros(16384 * 2, np.uint32)
    assert mod._lib.vhp_debug_read_units(C.c_void_p(u.ctypes.data), u.size) == 0
    u = u.reshape(-1, 2)[: 8 * n].astype(np.float64) * 0.01   # us
    t_in, t_out = u[:, 0], u[:, 1]
    geo = []
    for unit in range(8 * n):
        s, qo = divmod(unit, 8)
        sx, sy = int(src[s][0]), int(src[s][1])
        q = qo >> 1
        dx = 1 if q in (0, 3) else -1
        dy = 1 if q < 2 else -1
        ni = side - sx if dx > 0 else sx
        nj = side - sy if dy > 0 else sy
        xm = (qo & 1) == 0
        march = ni if xm else nj
        rows = min(ni, nj)
        cells = (rows * ni - rows * (rows - 1) / 2) if xm else max(min(ni, nj - 1), 0) * (nj - 1) - max(min(ni, nj - 1), 0) * (max(min(ni, nj - 1), 0) - 1) / 2
        geo.append((march, rows, cells, xm))
    geo = np.array(geo, float)
    live = t_out > 0
    print("%s buffer: launch %.3f ms; %d units with strips; last unit finishes at %.0f us" % (name, ms, int(live.sum()), t_out.max()))
    BIN = 50 if side >= 1000 else 10
    print("  installs per %d us:" % BIN, " ".join("%d" % ((t_in[live] >= a) & (t_in[live] < a + BIN)).sum() for a in range(0, 12 * BIN, BIN)))
    print("  finishes per %d us:" % BIN, " ".join("%d" % ((t_out[live] >= a) & (t_out[live] < a + BIN)).sum() for a in range(0, 12 * BIN, BIN)))
    print("  duration by march (x-major / y-major), us: median [p90]")
    for lo in range(0, side, max(side // 10, 1)):
        row = []
        for xm in (1.0, 0.0):
            sel = live & (geo[:, 0] >= lo) & (geo[:, 0] < lo + max(side // 10, 1)) & (geo[:, 3] == xm)
            d = (t_out - t_in)[sel]
            row.append("%6.0f [%6.0f] (%3d)" % (np.median(d), np.percentile(d, 90), sel.sum()) if sel.sum() else "     -")
        print("    march %4d-%4d: %s   %s" % (lo, lo + max(side // 10, 1), row[0], row[1]))
    last = np.argsort(-t_out)[:24]
    print("  the 24 units that finish last: unit, kind, march, rows, Mcells, installed at, finished at, duration")
    for k2 in last:
        print("    %5d %s march %4d rows %4d %6.3f Mcells  in %6.1f  out %6.1f  dur %6.1f" % (k2, "x" if geo[k2, 3] else "y", geo[k2, 0], geo[k2, 1], geo[k2, 2] / 1e6, t_in[k2], t_out[k2], t_out[k2] - t_in[k2]))
