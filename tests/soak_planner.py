"""Soak of the planner (not collected by pytest; run it on a GPU box: python3 tests/soak_planner.py [seconds] [seed]): random maps,
start / end points, thresholds and iteration caps; vhp_planner_solve against the oracle's solve() in every output, the speculative
solve's exact mode against both for k = 1, 2, 4, 8, and the fast mode's union, labels and local field against the oracle's sweeps of
the pivots it committed.  Statuses included: solved, max_iter, nothing lit."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import vhp_amd
import maps
from oracle_lib import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = Oracle()
t_end = time.time() + budget
n_cases = 0
stat = {}
KEYS = ("came_from", "vis_global", "vis_local")
while time.time() < t_end:
    nx, ny = int(rng.integers(9, 420)), int(rng.integers(9, 420))
    nb = int(rng.integers(2, 40))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 5, 2), 1, max(ny // 5, 2), seed=int(rng.integers(1 << 30)))
    free = np.argwhere(occ == 1)
    if len(free) < 2:
        continue
    a, b = free[rng.integers(len(free))], free[rng.integers(len(free))]
    start, end = (int(a[1]), int(a[0])), (int(b[1]), int(b[0]))
    thr = float(rng.choice([-0.5, 0.02, 0.1, 0.25, 0.5, 0.9, 1.0]))
    max_iter = int(rng.choice([1, 2, 7, 8, 9, 16, 40, 120]))
    ref = oracle.solve(occ, start, end, thr, max_iter)
    c = vhp_amd.Context(0)
    c.set_map(occ)
    got = c.planner_solve(start, end, thr, max_iter)
    def same(x, what):
        assert x["status"] == ref["status"], (what, x["status"], ref["status"], nx, ny, start, end, thr, max_iter)
        if ref["status"] not in (0, vhp_amd.VHP_ERR_MAX_ITER):
            return
        assert x["n_pivots"] == ref["n_pivots"] and np.array_equal(x["pivots"][: ref["n_pivots"] + 1], ref["pivots"]), (what, nx, ny, start, end, thr, max_iter)
        for k in KEYS:
            assert np.array_equal(x[k], ref[k]), (what, k, nx, ny, start, end, thr, max_iter)
    same(got, "plain")
    for k in (1, 2, 4, 8):
        same(c.planner_solve_speculative(start, end, thr, max_iter, k=k, mode=0), "exact k=%d" % k)
    k = int(rng.choice([2, 4, 8]))
    r = c.planner_solve_speculative(start, end, thr, max_iter, k=k, mode=1)
    if r["status"] in (0, vhp_amd.VHP_ERR_MAX_ITER) and thr >= 0:
        union = np.zeros(occ.shape)
        label = np.full(occ.shape, -1, np.int64)
        label[start[1], start[0]] = 0
        f = np.zeros(occ.shape)
        for j in range(r["n_pivots"]):
            f = oracle.sweep_full(occ, int(r["pivots"][j][0]), int(r["pivots"][j][1]))
            np.maximum(union, f, out=union)
            label[(label < 0) & (f >= thr)] = j
        lab = np.where(r["came_from"] == vhp_amd.UNLABELLED, -1, r["came_from"].astype(np.int64))
        assert np.array_equal(r["vis_global"], union), ("fast union", k, nx, ny, start, end, thr, max_iter)
        assert np.array_equal(lab, label), ("fast labels", k, nx, ny, start, end, thr, max_iter)
        assert np.array_equal(r["vis_local"], f), ("fast local", k, nx, ny, start, end, thr, max_iter)
    stat[ref["status"]] = stat.get(ref["status"], 0) + 1
    n_cases += 1
print("planner soak: %d cases, all equal; reference statuses %s" % (n_cases, stat))
