"""Per-workgroup timeline of one streaming-sweep launch (a -DVHP_EXP_WGTIME build): when each workgroup ran, on which
CU, how many slots it had and how busy each of its wavefronts was.  Diagnostic only.
usage: stream_timeline.py <lib built with -DVHP_EXP_WGTIME> [side] [n sources]"""
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
mod.LIB_PATH = os.path.join(ROOT, lib)
lo, hi = (20, 100) if side <= 1024 else (80, 400)
occ = synth.random_rect_map(side, side, 50, lo, hi, lo, hi, seed=1)
src = synth.free_sources(occ, n, seed=7)
if n == 1:  # the lone large quadrant
    occ[5, 5] = 1
    src = np.array([[5, 5]], np.int32)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", 2)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
for _ in range(5):
    c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
assert mod._lib.vhp_debug_clear_wgtime() == 0
c.timing(True)
c.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
print("kernel ms", c.timing_collect(4))
nu = 8 * n
buf = np.zeros(nu * 16, np.uint64)
rc = mod._lib.vhp_debug_read_wgtime(C.c_void_p(buf.ctypes.data), nu * 16)
assert rc == 0
w = buf.reshape(nu, 16)
t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
live = t1 > 0
base = t0[live].min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0  # us (100 MHz)
slots = (w[:, 2] & 0xffff).astype(int); ni = ((w[:, 2] >> 16) & 0xffff).astype(int); nj = ((w[:, 2] >> 32) & 0xffff).astype(int); octant = ((w[:, 2] >> 48) & 1).astype(int)
cyc = w[:, 4].astype(np.float64)
busy = w[:, 8:16].astype(np.float64)
print("workgroups live %d of %d; launch span %.1f us" % (live.sum(), nu, end[live].max()))
dur = end - start
print("start-time percentiles us:", np.percentile(start[live], [0, 25, 50, 75, 100]).round(1))
print("duration percentiles us  :", np.percentile(dur[live], [0, 25, 50, 75, 100]).round(1))
sel = live & (slots >= 10)
print("us per slot (median over WGs with >= 10 slots):", np.median((dur / np.maximum(slots, 1))[sel]).round(2))
frac = busy / np.maximum(cyc[:, None], 1)
for o, nm in ((0, "x-major units (waves 0-3 sweep, 4-7 flush)"), (1, "y-major units (waves 0-6 sweep)")):
    so = sel & (octant == o)
    if so.any():
        print("busy fraction per wavefront, %s: %s" % (nm, frac[so].mean(0).round(2)))
print("cycles per slot (median):", np.median((cyc / np.maximum(slots, 1))[sel]).round(0), " clock MHz ~", np.median((cyc / np.maximum(dur, 1e-3))[sel]).round(0))
ts = np.linspace(0, end[live].max(), 41)
conc = [(int(((start <= t) & (end > t) & live).sum())) for t in ts]
print("resident workgroups over time:", conc)
print("sum of slots", slots[live].sum(), " sum of WG-us", dur[live].sum().round(0))
for o, nm in ((0, "x-major"), (1, "y-major")):
    so = live & (octant == o)
    print("  %s units: %d, sum of WG-us %.0f, mean busy wavefronts per unit %.2f" % (nm, so.sum(), dur[so].sum(), (frac[so][:, :4 if o == 0 else 7].sum(1) * dur[so]).sum() / max(dur[so].sum(), 1e-9)))
big = np.argsort(-dur)[:8]
for b in big:
    print("  long unit %4d (%s): start %.1f dur %.1f us slots %d ni %d nj %d busy %s" % (b, "xy"[octant[b]], start[b], dur[b], slots[b], ni[b], nj[b], frac[b].round(2)))
np.save(os.path.join(ROOT, "gpurun_out", "stream_timeline.npy"), w)
# where the cycles of workgroup 0's wavefronts went (XWave / YWave prof[] of the diagnostic build)
pr = np.zeros(64, np.uint64)
if mod._lib.vhp_debug_read_prof(C.c_void_p(pr.ctypes.data)) == 0:
    pr = pr.reshape(8, 8).astype(np.float64)
    for wv in range(8):
        r = pr[wv]
        print("  a large unit's wave %d (%s): loads %.0f  steady windows %.0f  diag/pred windows %.0f  single steps %.0f  waits for the flusher %.0f  kcycles; %d windows -> %.0f cycles per window" % (
            wv, "x" if wv < 4 else "y", r[0] / 1e3, r[1] / 1e3, r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], (r[1] + r[2]) / max(r[5], 1)))
# who shared a CU with the largest units?
key = ((w[:, 3] >> 32) << 8) | ((w[:, 3] >> 8) & 0xff)
for b in big[:6]:
    same = np.where((key == key[b]) & live)[0]
    ov = [(int(u), round(float(max(0.0, min(end[u], end[b]) - max(start[u], start[b]))), 1), int(ni[u]) * int(nj[u]) // 1000) for u in same if u != b]
    ov = [o for o in ov if o[1] > 0]
    print("  unit %d (CU key %#x, %.0f us): %d other units overlapped it on that CU: %s" % (b, int(key[b]), dur[b], len(ov), ov[:12]))
