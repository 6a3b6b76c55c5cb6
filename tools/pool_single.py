"""Where one SINGLE-source pool-sweep launch spends its time (a -DVHP_DIAG_POOLPROF build).  Diagnostic only.
usage: pool_single.py <lib> [side] [contexts]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
lib = sys.argv[1]
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctxs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mod.LIB_PATH = os.path.join(ROOT, lib)
occ = np.ones((side, side), np.uint8)
src = np.array([[side // 2, side // 2]], np.int32)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
d_src = torch.from_numpy(src).cuda()
out = torch.empty((1, side, side), dtype=torch.float64, device="cuda")
for k in (1, 3, 4):
    c.set_option("kernel", k)
    if k == 3 and ctxs:
        c.set_option("pool_contexts", ctxs)
    for _ in range(5):
        c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
    torch.cuda.synchronize()
    c.timing(True)
    for _ in range(20):
        c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
    torch.cuda.synchronize()
    t = np.array(c.timing_collect(64))
    print("kernel %d: ms min %.4f median %.4f" % (k, t.min(), np.median(t)))
    c.timing(False)
if not hasattr(mod._lib, "vhp_debug_read_poolprof"):
    sys.exit(0)
W = 12
buf = np.zeros(512 * 16 * 12, np.uint64)
assert mod._lib.vhp_debug_read_poolprof(C.c_void_p(buf.ctypes.data), buf.size) == 0
w = buf.reshape(512, 16, 12)[:256, :W].astype(np.float64)
t0, t1 = w[..., 8], w[..., 9]
base = t0[t0 > 0].min()
busy = np.where(w[..., 5].sum(1) > 0)[0]
print("workgroups with strips:", busy.tolist())
print("all workgroups: start us min %.1f max %.1f, exit us min %.1f max %.1f" % (((t0 - base) / 100).min(), ((t0 - base) / 100).max(), ((t1 - base) / 100).min(), ((t1 - base) / 100).max()))
for g in busy:
    print("WG %3d" % g)
    for v in range(W):
        r = w[g, v]
        if r[5] == 0 and r[7] == 0:
            continue
        clk = r[10] / max((r[9] - r[8]) / 100.0, 1e-3)
        print("  wave %2d: start %6.1f exit %6.1f us | strips %2d blocks %3d installs %d | us: idle %6.1f fetch %6.1f sweep %6.1f install %6.1f blockload %5.1f | cyc/block %5.0f" % (
            v, (r[8] - base) / 100, (r[9] - base) / 100, r[5], r[6], r[7], r[0] / clk, r[1] / clk, r[2] / clk, r[3] / clk, r[4] / clk, r[2] / max(r[6], 1)))
ut = np.zeros(2 * 8, np.uint64)
if mod._lib.vhp_debug_read_unit_times(C.c_void_p(ut.ctypes.data), ut.size) == 0:
    ut = ut.reshape(-1, 2).astype(np.float64)
    for u in range(8):
        print("unit %d (%s-major): installed %.1f us, finished %.1f us" % (u, "xy"[u % 2], (ut[u, 0] - base) / 100, (ut[u, 1] - base) / 100))
