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
print("wave: steady windows n x cyc | diag windows n x cyc | slow steps n x cyc | barriers n x wait-cyc | busy us  wait us")
for w in range(16):
    a = v[w * 8: w * 8 + 8]
    f = lambda t, n: (n, t / n if n else 0)
    busy = (a[0] + a[2] + a[4]) / 2400.0
    print("%s%d: %3d x %5.0f | %3d x %5.0f | %3d x %5.0f | %3d x %6.0f | %6.1f %6.1f" % (("X" if w < 8 else "Y"), w % 8, *f(a[0], a[1]), *f(a[2], a[3]), *f(a[4], a[5]), *f(a[6], a[7]), busy, a[6] / 2400.0))
