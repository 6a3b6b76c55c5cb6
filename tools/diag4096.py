import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, vhp_amd, maps
from oracle_lib import Oracle
o = Oracle()
occ, src = maps.config_c5(128)
pick = np.concatenate([src[[3, 64]], np.array([[0, 0], [4095, 4095], [4095, 0], [1, 4094]], np.int32)])
pick = pick[[bool(occ[y, x]) for x, y in pick]]
c = vhp_amd.Context(0); c.set_map(occ)
for k,v in dict(kernel=1, rows_per_lane=1, strips=8, slide=1).items(): c.set_option(k, v)
got = c.sweep_batch(pick)
for k,(sx,sy) in enumerate(pick):
    want = o.sweep_full(occ, int(sx), int(sy))
    bad = np.argwhere(got[k] != want)
    if len(bad):
        ys, xs = bad[:,0], bad[:,1]
        print("src", sx, sy, "bad", len(bad), "x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "dx", xs.min()-sx, xs.max()-sx, "dy", ys.min()-sy, ys.max()-sy)
        print(bad[:70].tolist())
    else:
        print("src", sx, sy, "ok")
