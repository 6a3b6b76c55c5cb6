"""Bring-up helper: runs a few sweeps on the GPU and prints mismatch statistics against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import vhp_amd
import maps
from oracle_lib import Oracle

o = Oracle()
def run(occ, src, tag):
    c = vhp_amd.Context(0); c.set_map(occ)
    t = time.time(); got = c.sweep_batch(src); dt = time.time() - t
    nbad_total = 0
    for k, (sx, sy) in enumerate(src):
        want = o.sweep_full(occ, int(sx), int(sy))
        bad = np.argwhere(got[k] != want)
        nbad_total += len(bad)
        if len(bad):
            y, x = bad[0]
            dx, dy = bad[:, 1] - sx, bad[:, 0] - sy
            print("  %s src(%d,%d): %d bad; first (x=%d,y=%d) got %r want %r; dx range [%d,%d] dy range [%d,%d] maxerr %g" % (
                tag, sx, sy, len(bad), x, y, got[k][y, x], want[y, x], dx.min(), dx.max(), dy.min(), dy.max(),
                np.nanmax(np.abs(got[k] - want))))
    print("%s: %d sources, %d bad cells, %.3fs" % (tag, len(src), nbad_total, dt))

print(vhp_amd.version(), torch.cuda.get_device_name(0))
run(np.ones((9, 9), np.uint8), np.array([[2, 2], [0, 0], [8, 8]], np.int32), "9x9 empty")
occ = np.ones((9, 9), np.uint8); occ[3, 4] = 0
run(occ, np.array([[2, 2]], np.int32), "9x9 Q1 probe")
occ = maps.random_rect_map(101, 77, 25, 2, 20, 2, 20, 1)
run(occ, maps.free_sources(occ, 8, 5), "101x77")
occ = maps.random_rect_map(300, 263, 30, 3, 40, 3, 40, 2)
run(occ, maps.free_sources(occ, 8, 5), "300x263")
occ, src = maps.config_c3(8)
run(occ, src, "C3 1000x1000")
