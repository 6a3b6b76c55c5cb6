"""BASELINE config 1 (101 x 101 rnd_1 mask) and two more odd-width maps through the planner's loop: the latency sweep (taken by
itself since round 4) against the front sweep (vhp_set_option "kernel" 1), device loop time per pivot.  Diagnostic only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import vhp_amd, maps
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
cases = [("C1 rnd_1 mask 101x101, (5,5) -> (95,95), thr 0.5", maps.c1_rnd1_mask(), (5, 5), (95, 95), 0.5)]
for side in (501, 1001):
    occ = synth.random_rect_map(side, side, 50, side // 50, side // 10, side // 50, side // 10, seed=1)
    occ[5, 5] = occ[side - 10, side - 10] = 1
    cases.append(("random %dx%d, (5,5) -> (%d,%d), thr 0.25" % (side, side, side - 10, side - 10), occ, (5, 5), (side - 10, side - 10), 0.25))
for name, occ, s, e, thr in cases:
    res = {}
    for k in (1, 0):
        c = vhp_amd.Context(0)
        c.set_map(occ)
        c.set_option("kernel", k)
        ms = []
        for it in range(6):
            rc, npiv, _ = c.planner_solve_device(s, e, thr, 250)
            if it:
                ms.append(c.last_elapsed_ms())
        res[k] = (float(np.median(ms)), npiv, rc, c.last_sweep_kernel())
    (a, n, rc, ka), (b, _, _, kb) = res[1], res[0]
    print("%-52s %3d pivots (status %d): front sweep (kernel %d) %7.1f us/pivot, default (kernel %d) %7.1f us/pivot  (%.2f)" % (name, n, rc, ka, 1e3 * a / max(n, 1), kb, 1e3 * b / max(n, 1), b / a))
