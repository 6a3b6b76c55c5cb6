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
    print("pre %.0f rows %.0f ringwr %.0f per step |" % (v[6]/500, v[7]/500, v[8]/500), end=" ")
    print("init %d | slots-work %d | flush-part %d | barrier-wait %d | total %d | steps %d  -> per step: work %.0f flush %.0f; total us %.1f" % (
        v[0], v[1], v[2], v[3], v[4], v[5], (v[1]) / max(v[5], 1), v[2] / max(v[5], 1), v[4] / 2400.0))
