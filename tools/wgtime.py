"""Per-workgroup start/end times of one sweep launch (diagnostic build exp/libvhp_WGTIME.so, which dumps them to
gpurun_out/wgtime.bin after every launch).  Shows how the launch's makespan is composed.
usage: VHP_LIB=exp/libvhp_WGTIME.so python tools/wgtime.py [n_sources]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
side = 1000
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
out = torch.empty((n, side, side), dtype=torch.float64, device="cuda")
for _ in range(5):
    ctx.sweep_batch_device(d_src.data_ptr(), n, out.data_ptr(), dtype=vhp_amd.F64)
torch.cuda.synchronize()
raw = np.fromfile(os.path.join(ROOT, "gpurun_out", "wgtime.bin"), dtype=np.int64).reshape(n, 4, 3)
t0, t1, hw = raw[..., 0].astype(np.float64), raw[..., 1].astype(np.float64), raw[..., 2]
base = t0.min()
t0 = (t0 - base) / 100.0  # us
t1 = (t1 - base) / 100.0
dur = t1 - t0
ni = np.where(np.array([[1, 0, 0, 1]]) == 1, side - src[:, :1], src[:, :1])
nj = np.where(np.array([[1, 1, 0, 0]]) == 1, side - src[:, 1:2], src[:, 1:2])
L = np.maximum(ni, nj)
mn = np.minimum(ni, nj)
print("launch makespan %.1f us; workgroups %d" % (t1.max(), dur.size))
print("start times: %d start within 5 us; median start %.1f us; last start %.1f us" % ((t0 < 5).sum(), np.median(t0), t0.max()))
order = np.argsort(t1.ravel())[::-1]
print("last to finish: (end us, start us, dur us, ni, nj)")
for k in order[:8]:
    print("   %.1f %.1f %.1f  %d %d" % (t1.ravel()[k], t0.ravel()[k], dur.ravel()[k], ni.ravel()[k], nj.ravel()[k]))
for lo, hi in [(0, 250), (250, 500), (500, 750), (750, 1001)]:
    sel = (L >= lo) & (L < hi)
    if sel.any():
        print("front length %4d..%4d: %4d units, duration mean %.1f us (min %.1f max %.1f), us per 8 columns %.2f" %
              (lo, hi, sel.sum(), dur[sel].mean(), dur[sel].min(), dur[sel].max(), (dur[sel] / (L[sel] / 8.0 + 8)).mean()))
# utilisation over time: busy workgroup slots
edges = np.linspace(0, t1.max(), 21)
busy = [(np.minimum(t1, b) - np.maximum(t0, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups over time (20 bins):", " ".join("%d" % x for x in busy))
