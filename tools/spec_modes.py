"""The planner's plain loop and the speculative modes on maze_6, device loop only: ms per solve, launches, hits.  Diagnostic only.
usage: spec_modes.py [reps] [threshold] [max_iter]     (threshold 0.25 on maze_6: the pivot repeats until max_iter; under rocprofv3 --kernel-trace: tools/spec_trace_split.py splits the trace per configuration)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 250
occ = import_module("visibility-heuristic-path-planner_amd.synth").maze_6()
ny = occ.shape[0]
start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)   # (bench.py's C4: mode 2, y flipped)
ctx = mod.Context(0)
ctx.set_map(occ)
ms = []
for it in range(reps + 1):
    rc, n_piv, _ = ctx.planner_solve_device(start, end, thr, max_iter)
    if it: ms.append(ctx.last_elapsed_ms())
print("threshold %g, max_iter %d" % (thr, max_iter))
print("plain: status %d, %d pivots, %.3f ms per solve, %.2f us per pivot" % (rc, n_piv, np.mean(ms), np.mean(ms) * 1e3 / max(n_piv, 1)))
for mode in (0, 1):
    for k in (1, 2, 4, 8):
        ms = []
        for it in range(reps + 1):
            r = ctx.planner_solve_speculative(start, end, thr, max_iter, k=k, mode=mode, outputs=False)
            if it: ms.append(ctx.last_elapsed_ms())
        it_n = r["sweeps"] + r["hits"]
        print("%s k=%d: status %d, %d pivots, %d iterations (%d hits), %.3f ms per solve, %.2f us per iteration" %
              ("exact" if mode == 0 else "fast", k, r["status"], r["n_pivots"], it_n, r["hits"], np.mean(ms), np.mean(ms) * 1e3 / it_n))
