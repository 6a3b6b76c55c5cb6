"""Per-slot barrier timeline of a single-source sweep (diagnostic build exp/libvhp_SLOTTIME.so): for each pipeline
slot, when each wavefront reached the slot's barrier and when it left it.  Shows which wavefront a slot waits for.
usage: VHP_LIB=exp/libvhp_SLOTTIME.so python tools/slottime.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vhp_amd
occ = np.ones((1000, 1000), np.uint8)
src = np.array([[500, 500]], np.int32)
ctx = vhp_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_map(occ)
d_src = torch.from_numpy(src).cuda()
out = torch.empty((1, 1000, 1000), dtype=torch.float64, device="cuda")
for _ in range(5):
    ctx.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr(), dtype=vhp_amd.F64)
torch.cuda.synchronize()
raw = np.fromfile(os.path.join(ROOT, "gpurun_out", "slottime.bin"), dtype=np.int64).reshape(4, 16, 160, 2).astype(np.float64)
for q in range(4):
    a, b = raw[q, :, :, 0], raw[q, :, :, 1]
    nslots = int((a[0] > 0).sum())
    a, b = a[:, :nslots], b[:, :nslots]
    t0 = a.min()
    a, b = (a - t0) * 10.0, (b - t0) * 10.0  # ns
    rel = b.max(axis=0)              # slot end = last wave leaves the barrier
    dur = np.diff(np.concatenate([[0.0], rel]))
    last = a.argmax(axis=0)          # wavefront that arrives last
    wait = (b - a)                   # time spent at the barrier
    print("quadrant %d: %d slots, total %.1f us, mean slot %.0f ns" % (q, nslots, rel[-1] / 1000, dur.mean()))
    print("  last-arriving wavefront per slot:", " ".join("%d" % w for w in last))
    print("  slot durations (ns):", " ".join("%d" % d for d in dur))
    print("  barrier exit - last arrival (ns):", " ".join("%d" % x for x in (b.max(axis=0) - a.max(axis=0))))
    busy = 1.0 - wait.sum(axis=1) / rel[-1]
    print("  busy fraction per wavefront:", " ".join("%.2f" % x for x in busy))
    # work time of a wavefront in a slot = arrival at this slot's barrier - exit from the previous one
    work = a - np.concatenate([np.zeros((16, 1)), b[:, :-1]], axis=1)
    for wv in (0, 1, 3, 8, 9, 11):
        print("  wavefront %2d work per slot (ns):" % wv, " ".join("%d" % x for x in work[wv]))
