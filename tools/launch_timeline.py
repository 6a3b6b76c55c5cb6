"""Bytes swept and strips running per 10 us of one pool-sweep launch (a -DVHP_DIAG_POOLPROF build).  Diagnostic only.
usage: launch_timeline.py <lib> <tag> [n sources] [n candidate buffers]
The launch is timed on several hipMalloc'ed outputs; the timeline is taken on the slowest and on the fastest of them."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
lib, tag = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
nbuf = int(sys.argv[4]) if len(sys.argv) > 4 else 12
side = 1000
mod.LIB_PATH = os.path.join(ROOT, lib)
occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
src = synth.free_sources(occ, n, seed=7)
c = mod.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_map(occ)
c.set_option("kernel", 3)
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
times = [run(p, 6) for p in bufs]
print(tag, "launch ms per candidate buffer:", " ".join("%.3f" % t for t in times), flush=True)
order = np.argsort(times)
picks = [("slowest", int(order[-1])), ("fastest", int(order[0]))]
rows = {}
for name, k in picks:
    ms = run(bufs[k], 1)
    h = np.zeros(256 * 2 * 256, np.uint64)
    assert mod._lib.vhp_debug_read_hist(C.c_void_p(h.ctypes.data), h.size) == 0
    h = h.reshape(256, 2, 256)
    by = h[:, 0].astype(np.float64).sum(0)
    st = np.cumsum(h[:, 1].astype(np.int64).sum(0))
    last = int(np.nonzero(by)[0].max()) + 1
    print("%s %s buffer %d: %.3f ms; bytes in the histogram %.3f GB over %d bins of 10 us" % (tag, name, k, ms, by.sum() / 1e9, last), flush=True)
    # per workgroup: wavefronts sweeping a strip at the end of every bin -- how many workgroups have (nearly) all of theirs busy, i.e.
    # could use a wavefront of another CU, and how many have idle ones
    per_wg = np.cumsum(h[:, 1].astype(np.int64), axis=1)
    wmax = int(per_wg.max())
    print("  t_us   GB/s(10us)  strips running   workgroups with >= %d sweeping   with <= %d" % (wmax - 1, wmax // 2))
    for b in range(last):
        print("  %4d   %8.0f   %5d   %5d   %5d" % (10 * b, by[b] / 1e-5 / 1e9, st[b], int((per_wg[:, b] >= wmax - 1).sum()), int((per_wg[:, b] <= wmax // 2).sum())))
    # per workgroup: the bytes it swept in all and the bin in which it swept its last -- is the end of the launch a few workgroups with
    # more than their share, or all of them running out together?
    wg_bytes = h[:, 0].astype(np.float64).sum(1) / 1e6
    wg_end = np.array([int(np.nonzero(h[g, 0])[0].max()) if h[g, 0].any() else 0 for g in range(256)]) * 10
    print("  per workgroup: MB swept min %.2f / mean %.2f / max %.2f; last bin with bytes (us) min %d / median %d / p90 %d / max %d" % (
        wg_bytes.min(), wg_bytes.mean(), wg_bytes.max(), wg_end.min(), int(np.median(wg_end)), int(np.percentile(wg_end, 90)), wg_end.max()))
    print("  correlation (MB swept, end time) %.2f; the 8 workgroups that end last: %s" % (
        float(np.corrcoef(wg_bytes, wg_end)[0, 1]), " ".join("%d:%.1fMB@%dus" % (g, wg_bytes[g], wg_end[g]) for g in np.argsort(-wg_end)[:8])))
    q = np.argsort(wg_end)
    print("  mean MB of the 64 workgroups that end first %.2f, of the 64 that end last %.2f" % (wg_bytes[q[:64]].mean(), wg_bytes[q[-64:]].mean()))
    rows[name] = (by[:last], st[:last], ms)
with open(os.path.join(ROOT, "gpurun_out", "timeline_%s.csv" % tag), "w") as f:
    f.write("# %s: pool sweep, %d sources at %d^2; bytes swept per 10 us bin (GB/s) and strips running; launch ms: %s\n" % (tag, n, side, ", ".join("%s %.3f" % (k, v[2]) for k, v in rows.items())))
    f.write("t_us," + ",".join("%s_GBps,%s_strips" % (k, k) for k in rows) + "\n")
    L = max(len(v[0]) for v in rows.values())
    for b in range(L):
        f.write("%d," % (10 * b) + ",".join(("%.0f,%d" % (v[0][b] / 1e-5 / 1e9, v[1][b])) if b < len(v[0]) else "," for v in rows.values()) + "\n")
