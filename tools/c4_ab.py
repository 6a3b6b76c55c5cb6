"""The planner's plain loop on maze_6 (BASELINE config 4) under several library builds, one process: us per pivot of the device loop.
usage: c4_ab.py <lib> ...   ("-" = the in-tree build).  Diagnostic only."""
import os, sys, ctypes as C
ROOT = "/root/repo" if os.path.exists("/root/repo") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np, torch
from importlib import import_module
mod = import_module("visibility-heuristic-path-planner_amd")
synth = import_module("visibility-heuristic-path-planner_amd.synth")
occ = synth.maze_6()
for lib in sys.argv[1:]:
    mod._lib = None
    mod.LIB_PATH = os.path.join(mod._HERE, "libvhp_hip.so") if lib == "-" else os.path.join(ROOT, lib)
    c = mod.Context(0); c.set_map(occ)
    ts = []
    for rep in range(40):
        ny = occ.shape[0]
        r = c.planner_solve_device((345, ny - 1 - 391), (341, ny - 1 - 10), 0.1, 250)   # (BASELINE config 4: mode 2 flips y)
        ts.append(c.last_elapsed_ms())
    print(lib, "status %d, %d pivots: device loop ms median %.4f -> %.2f us per pivot" % (r[0], r[1], np.median(ts[2:]), 1e3 * np.median(ts[2:]) / max(r[1], 1)))
