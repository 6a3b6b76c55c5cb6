import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import vhp_amd
lib = vhp_amd.load_library()
ctx = vhp_amd.Context(0)
occ = np.ones((1000, 1000), np.uint8); ctx.set_map(occ)
src = np.array([[500, 500]], np.int32)
buf = (ctypes.c_ulonglong * 256)()
ctx.sweep_batch(src); lib.vhp_debug_fetch(buf, 1)
ctx.sweep_batch(src); lib.vhp_debug_fetch(buf, 1)
v = list(buf)
for name, o in (("X0", 0), ("Y0", 16)):
    for kind, b in (("steady", 0), ("diag", 4)):
        n = max(v[o + b + 3], 1)
        print("%s %-6s windows %3d: preamble %5.0f  8 steps %5.0f  flush %5.0f" % (name, kind, v[o + b + 3], v[o + b] / n, v[o + b + 1] / n, v[o + b + 2] / n))
