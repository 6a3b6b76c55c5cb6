"""When the strips of one latency-sweep launch (C2: one source, centre of an empty grid) were set up, started, finished their
first block and ended (a -DVHP_DIAG_POOLPROF build).  Diagnostic only.
usage: lat_timeline.py <lib> [side] [sx sy]      (side = 0: maze_6, the planner's case -- most strips die a few windows in)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
lib = sys.argv[1]
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sx = int(sys.argv[3]) if len(sys.argv) > 4 else side // 2
sy = int(sys.argv[4]) if len(sys.argv) > 4 else side // 2
mod.LIB_PATH = os.path.join(ROOT, lib)
occ = np.ones((side, side), np.uint8)
if side == 0:
    occ = import_module("visibility-heuristic-path-planner_amd.synth").maze_6()
ny_, nx_ = occ.shape
src = np.array([[sx, sy]], np.int32)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
d_src = torch.from_numpy(src).cuda()
out = torch.empty((1, ny_, nx_), dtype=torch.float64, device="cuda")
c.set_option("kernel", 4)
for _ in range(5):
    c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
torch.cuda.synchronize()
c.timing(True)
for _ in range(20):
    c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
torch.cuda.synchronize()
t = np.array(c.timing_collect(64))
print("latency sweep: ms min %.4f median %.4f" % (t.min(), np.median(t)))
buf = np.zeros(256 * 16 * 20, np.uint64)
assert mod._lib.vhp_debug_read_latprof(C.c_void_p(buf.ctypes.data), buf.size) == 0
wall = buf.reshape(256, 16, 20)[:8].astype(np.float64)
w = wall[:, :12]
base = w[..., 16][w[..., 16] > 0].min()
st = np.zeros(64 * 48 * 4, np.uint64)
assert mod._lib.vhp_debug_read_lat_strip_times(C.c_void_p(st.ctypes.data), st.size) == 0
st = st.reshape(64, 48, 4).astype(np.float64)
for u in range(8):
    print("unit %d (%s-major): workgroup start %.1f us, last wavefront out %.1f us" % (u, "xy"[u % 2], (w[u, :, 16].min() - base) / 100, (w[u, :, 17].max() - base) / 100))
    if True:
        for v in range(12):
            r = w[u, v]
            nw = r[12] + r[13]
            if nw > 0 and r[12] == 0:   # the sweep in bands: one kind of window
                print("   wave %2d: %3d windows of %5.0f cycles: operands %5.0f compute %5.0f (of it waiting for the storer %5.0f) request, publish, post %5.0f | asked the band below again %d times" % (
                    v, r[13], r[14] / max(r[13], 1), r[8] / max(r[13], 1), r[9] / max(r[13], 1), r[11] / max(r[13], 1), r[10] / max(r[13], 1), r[7]))
                rs = wall[u, v + 8]
                if rs[12] > 0:
                    print("      its storer: %3d records, %5.0f cycles each from record to last store, %5.0f waiting for the next record" % (rs[12], rs[15] / rs[12], rs[6] / rs[12]))
            elif nw > 0:
                print("   wave %2d: %3d steady windows of %5.0f cycles, %2d diagonal of %5.0f; diagonal windows: operands %5.0f compute %5.0f tile reads + next request %5.0f stores %5.0f | whole strip %7.0f cycles, outside the windows %6.0f; asked the strip below again %d times" % (
                    v, r[12], r[15] / max(r[12], 1), r[13], r[14] / max(r[13], 1), r[8] / max(r[13], 1), r[9] / max(r[13], 1), r[10] / max(r[13], 1), r[11] / max(r[13], 1), r[2], r[2] - r[15] - r[14], r[7]))
    for p in range(48):
        if st[u, p, 3] == 0:
            continue
        r = (st[u, p] - base) / 100
        name = "diag  " if p == 47 else "strip %d" % p
        print("   %s: set up %6.1f  started %6.1f  first block done %6.1f (%5.1f)  ended %6.1f" % (name, r[0], r[1], r[2], r[2] - r[1], r[3]))
