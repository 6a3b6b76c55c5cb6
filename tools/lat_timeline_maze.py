"""lat_timeline.py on the planner's maze (690 x 402) from one of its pivots: when strips start, die and end.  Diagnostic only.
usage: lat_timeline_maze.py <lib built with -DVHP_DIAG_POOLPROF> [sx sy]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
mod.LIB_PATH = os.path.join(ROOT, sys.argv[1])
occ = synth.maze_6()
ny, nx = occ.shape
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (345, 391)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", 4)
d_src = torch.from_numpy(np.array([[sx, sy]], np.int32)).cuda()
out = torch.empty((1, ny, nx), dtype=torch.float64, device="cuda")
for _ in range(5):
    c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
torch.cuda.synchronize()
c.timing(True)
for _ in range(20):
    c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
torch.cuda.synchronize()
t = np.array(c.timing_collect(64))
print("maze_6 from (%d,%d): latency sweep ms min %.4f median %.4f" % (sx, sy, t.min(), np.median(t)))
buf = np.zeros(256 * 16 * 20, np.uint64)
assert mod._lib.vhp_debug_read_latprof(C.c_void_p(buf.ctypes.data), buf.size) == 0
w = buf.reshape(256, 16, 20)[:8, :12].astype(np.float64)
base = w[..., 16][w[..., 16] > 0].min()
st = np.zeros(64 * 48 * 4, np.uint64)
assert mod._lib.vhp_debug_read_lat_strip_times(C.c_void_p(st.ctypes.data), st.size) == 0
st = st.reshape(64, 48, 4).astype(np.float64)
for u in range(8):
    print("unit %d (%s-major): last wavefront out %.1f us" % (u, "xy"[u % 2], (w[u, :, 17].max() - base) / 100))
    for p in range(48):
        if st[u, p, 3] == 0:
            continue
        r = (st[u, p] - base) / 100
        print("   %s: set up %6.1f  started %6.1f  ended %6.1f" % ("diag  " if p == 47 else "strip %d" % p, r[0], r[1], r[3]))
