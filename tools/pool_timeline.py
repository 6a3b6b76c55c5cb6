"""Where the wavefronts of one pool-sweep launch spent their cycles (a -DVHP_EXP_POOLPROF build).  Diagnostic only.
usage: pool_timeline.py <lib built with -DVHP_EXP_POOLPROF> [side] [n sources] [contexts]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
lib = sys.argv[1]
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ctxs = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mod.LIB_PATH = os.path.join(ROOT, lib)
lo, hi = (20, 100) if side <= 1024 else (80, 400)
occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
src = synth.free_sources(occ, n, seed=7)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", 3)
if ctxs:
    c.set_option("pool_contexts", ctxs)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
for _ in range(5):
    c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
c.timing(True)
c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
print("kernel ms", c.timing_collect(4))
W = 12
buf = np.zeros(512 * 16 * 12, np.uint64)
assert mod._lib.vhp_debug_read_poolprof(C.c_void_p(buf.ctypes.data), buf.size) == 0
w = buf.reshape(512, 16, 12)[:256, :W].astype(np.float64)
t0, t1 = w[..., 8], w[..., 9]
base = t0.min()
end_us = (t1 - base) / 100.0
start_us = (t0 - base) / 100.0
cyc = w[..., 10]
print("workgroup start us: min %.1f max %.1f;  wavefront exit us: percentiles %s" % (start_us.min(), start_us.max(), np.percentile(end_us, [0, 10, 50, 90, 100]).round(1)))
wg_end = end_us.max(1)
print("workgroup end us percentiles:", np.percentile(wg_end, [0, 10, 25, 50, 75, 90, 100]).round(1))
names = ["idle/looking", "  of the sweep: boundary fetch", "sweep (+stores)", "install (+diag)", "  of the sweep: block-start loads"]
tot = cyc.sum()
for k, nm in enumerate(names):
    print("  %-34s %5.1f %% of wavefront cycles" % (nm, 100 * w[..., k].sum() / tot))
print("  (unaccounted        %5.1f %%)" % (100 * (1 - (w[..., 0] + w[..., 2] + w[..., 3]).sum() / tot)))
print("strips per wavefront: mean %.1f; blocks %.1f; units installed per WG %.1f" % (w[..., 5].mean(), w[..., 6].mean(), w[..., 7].sum(1).mean()))
print("cycles per block swept (sweep cycles / blocks): %.0f;  clock MHz ~ %.0f" % (w[..., 2].sum() / w[..., 6].sum(), np.median(cyc / np.maximum((t1 - t0) / 100.0, 1e-3))))
# the time after which the chip runs dry: busy wavefronts over time cannot be reconstructed from sums, but the exit times can
ts = np.linspace(0, end_us.max(), 21)
print("wavefronts still running at t:", [(round(float(t), 0), int((end_us > t).sum())) for t in ts])
np.save(os.path.join(ROOT, "gpurun_out", "pool_timeline.npy"), w)
ut = np.zeros(2 * 8 * n, np.uint64)
if mod._lib.vhp_debug_read_unit_times(C.c_void_p(ut.ctypes.data), ut.size) == 0:
    ut = ut.reshape(-1, 2).astype(np.float64)
    ok = ut[:, 1] > 0
    t_in, t_out = (ut[:, 0] - base) / 100.0, (ut[:, 1] - base) / 100.0
    def geo(u):
        s_, qo = divmod(u, 8)
        q = qo >> 1
        sx, sy = int(src[s_][0]), int(src[s_][1])
        ni = side - sx if q in (0, 3) else sx
        nj = side - sy if q < 2 else sy
        return ("x" if qo % 2 == 0 else "y"), ni, nj
    print("units: installed at (percentiles) %s us; finished at %s us; lifetime %s us" % (np.percentile(t_in[ok], [0, 25, 50, 75, 100]).round(0),
          np.percentile(t_out[ok], [0, 25, 50, 75, 100]).round(0), np.percentile((t_out - t_in)[ok], [0, 25, 50, 75, 100]).round(0)))
    last = np.argsort(-np.where(ok, t_out, 0))[:14]
    for u in last:
        k, ni, nj = geo(int(u))
        print("  late unit %5d (%s-major, ni %4d nj %4d): installed %.0f us, finished %.0f us" % (u, k, ni, nj, t_in[u], t_out[u]))
    big = [u for u in range(8 * n) if ok[u] and min(geo(u)[1:]) > 0.8 * side][:10]
    for u in big:
        k, ni, nj = geo(int(u))
        print("  big unit %5d (%s-major, ni %4d nj %4d): installed %.0f us, finished %.0f us" % (u, k, ni, nj, t_in[u], t_out[u]))
