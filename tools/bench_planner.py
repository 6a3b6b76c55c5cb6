"""C4: full visibility-heuristic planner on maze_6 (690x402, thr 0.1), GPU vs the CPU oracle on this host."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import vhp_amd
from importlib import import_module
synth = import_module("visibility-heuristic-path-planner_amd.synth")
from oracle_lib import Oracle
occ = synth.maze_6(); ny = occ.shape[0]
start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
ctx = vhp_amd.Context(0); ctx.set_map(occ)
r = ctx.planner_solve(start, end, 0.1, 250)
ts = []
for _ in range(5):
    t = time.perf_counter(); r = ctx.planner_solve(start, end, 0.1, 250); ts.append(time.perf_counter() - t)
kern_ms = ctx.last_elapsed_ms()
o = Oracle()
t = time.perf_counter(); w = o.solve(occ, start, end, 0.1, 250); tc = time.perf_counter() - t
same = r["pivots"].tolist() == w["pivots"].tolist() and r["came_from"].tobytes() == w["came_from"].tobytes()
print("maze_6: %d pivots, parity %s | GPU solve %.2f ms wall (min of 5, incl. 5.5 MB of D2H results), %.2f ms device loop -> %.0f pivots/s | CPU oracle %.1f ms -> %.0f pivots/s" % (
    r["n_pivots"], same, min(ts) * 1e3, kern_ms, r["n_pivots"] / (kern_ms * 1e-3), tc * 1e3, w["n_pivots"] / tc))
occ, _ = synth.config_c3(1)
ctx.set_map(occ)
r = ctx.planner_solve((5, 5), (990, 990), 0.25, 250)
t = time.perf_counter(); r = ctx.planner_solve((5, 5), (990, 990), 0.25, 250); tg = time.perf_counter() - t
t = time.perf_counter(); w = o.solve(occ, (5, 5), (990, 990), 0.25, 250); tc = time.perf_counter() - t
print("1000^2 random map: %d pivots, parity %s | GPU %.2f ms (device loop %.2f ms) | CPU oracle %.1f ms" % (
    r["n_pivots"], r["pivots"].tolist() == w["pivots"].tolist(), tg * 1e3, ctx.last_elapsed_ms(), tc * 1e3))
