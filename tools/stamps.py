import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import vhp_amd
lib = vhp_amd.load_library()
ctx = vhp_amd.Context(0)
occ = np.ones((1000, 1000), np.uint8); ctx.set_map(occ)
src = np.array([[500, 500]], np.int32)
buf = (ctypes.c_ulonglong * 32)()
for rep in range(3):
    ctx.sweep_batch(src)
    lib.vhp_debug_fetch(buf, 1)
    v = list(buf)
    nw = max(v[3], 1)
    print("fast windows %d: preamble %.0f  steps(8) %.0f  flush %.0f cycles each | slow steps %d: %.0f each | slots %d: work %.0f barrier-wait %.0f each | kernel-loop total %.1f us" % (
        v[3], v[0] / nw, v[1] / nw, v[2] / nw, v[7], v[6] / max(v[7], 1), v[9], v[4] / max(v[9], 1), v[5] / max(v[9], 1), v[8] / 2400.0))
