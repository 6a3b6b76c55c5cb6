"""The speculative planner (vhp_planner_solve_speculative): every sweep launch takes the k - 1 best other
candidates along.  mode 0 must equal the plain planner -- hence the oracle -- in every output, whatever k; mode 1 commits
all candidates and must still produce a valid parent table and a path from the end to the start."""
import numpy as np
import pytest

import maps
from test_gpu_planner import _assert_same_solution

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def _ctx(vhp, occ):
    c = vhp.Context(0)
    c.set_map(occ)
    return c


@pytest.mark.parametrize("k", [1, 2, 4, 8])
def test_exact_mode_equals_the_oracle_on_maze6(vhp, oracle, k):
    occ = maps.maze_6()
    ny = occ.shape[0]
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    c = _ctx(vhp, occ)
    got = c.planner_solve_speculative(start, end, 0.1, 250, k=k, mode=0)
    want = oracle.solve(occ, start, end, 0.1, 250)
    _assert_same_solution(got, want, "maze_6, k=%d" % k)
    assert got["n_pivots"] == 64 and got["hits"] + got["sweeps"] == 64
    plain = c.planner_solve(start, end, 0.1, 250)
    _assert_same_solution(got, plain, "maze_6 vs vhp_planner_solve, k=%d" % k)
    if k == 1:
        assert got["fields_swept"] == got["sweeps"]


@pytest.mark.parametrize("seed,thr,k", [(1, 0.25, 4), (2, 0.5, 8), (3, 0.1, 2), (4, 0.25, 8), (5, 0.9, 4)])
def test_exact_mode_random_maps(vhp, oracle, seed, thr, k):
    occ = maps.random_rect_map(160, 131, 22, 4, 30, 4, 30, seed)
    pts = maps.free_sources(occ, 2, seed + 50)
    start, end = tuple(int(v) for v in pts[0]), tuple(int(v) for v in pts[1])
    got = _ctx(vhp, occ).planner_solve_speculative(start, end, thr, 60, k=k, mode=0)
    _assert_same_solution(got, oracle.solve(occ, start, end, thr, 60), "seed %d thr %g k %d" % (seed, thr, k))


def test_exact_mode_c1_mask_and_1000(vhp, oracle):
    occ = maps.c1_rnd1_mask()
    got = _ctx(vhp, occ).planner_solve_speculative((5, 5), (95, 95), 0.3, 100, k=4, mode=0)
    _assert_same_solution(got, oracle.solve(occ, (5, 5), (95, 95), 0.3, 100), "C1 mask")
    occ, _ = maps.config_c3(4)
    occ = occ.copy()
    occ[50, 50] = occ[990, 990] = 1
    got = _ctx(vhp, occ).planner_solve_speculative((50, 50), (990, 990), 0.25, 250, k=8, mode=0)
    _assert_same_solution(got, oracle.solve(occ, (50, 50), (990, 990), 0.25, 250), "1000^2")


def test_exact_mode_livelock_always_hits(vhp, oracle):
    # thr 0.25 on maze_6 repeats one pivot until max_iter: after its first sweep that pivot's field is cached
    occ = maps.maze_6()
    ny = occ.shape[0]
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    got = _ctx(vhp, occ).planner_solve_speculative(start, end, 0.25, 60, k=2, mode=0)
    want = oracle.solve(occ, start, end, 0.25, 60)
    _assert_same_solution(got, want, "live-lock")
    assert got["status"] == vhp.VHP_ERR_MAX_ITER
    repeats = sum(1 for a, b in zip(want["pivots"][:-1], want["pivots"][1:]) if tuple(a) == tuple(b))
    assert got["hits"] >= repeats - 1 > 10


def test_validation_codes_and_arguments(vhp):
    occ = np.ones((64, 64), np.uint8)
    occ[2, 2] = 0
    c = _ctx(vhp, occ)
    assert c.planner_solve_speculative((70, 1), (1, 1), 0.5, 5)["status"] == vhp.VHP_ERR_START_OOB
    assert c.planner_solve_speculative((1, 1), (1, 64), 0.5, 5)["status"] == vhp.VHP_ERR_END_OOB
    assert c.planner_solve_speculative((2, 2), (1, 1), 0.5, 5)["status"] == vhp.VHP_ERR_START_OCCUPIED
    assert c.planner_solve_speculative((1, 1), (2, 2), 0.5, 5)["status"] == vhp.VHP_ERR_END_OCCUPIED
    with pytest.raises(vhp.VhpError):
        c.planner_solve_speculative((1, 1), (60, 60), 0.5, 5, k=3)
    with pytest.raises(vhp.VhpError):
        c.planner_solve_speculative((1, 1), (60, 60), 0.5, 5, mode=2)


def _check_valid_plan(vhp, occ, r, start, end, thr):
    assert r["status"] == vhp.VHP_OK
    piv, came, vg = r["pivots"], r["came_from"], r["vis_global"]
    n = r["n_pivots"]
    assert tuple(piv[0]) == tuple(start) and tuple(piv[n]) == tuple(end)
    assert vg[end[1], end[0]] > thr
    lab = came[came != vhp.UNLABELLED]
    assert lab.max() < n
    # every pivot but the first was lit by an earlier pivot
    for k in range(1, n):
        x, y = int(piv[k][0]), int(piv[k][1])
        assert came[y, x] < k, "pivot %d at (%d,%d) carries label %d" % (k, x, y, came[y, x])
        assert occ[y, x] == 1
    d, path = vhp.reconstruct_path(came, piv, end)
    assert tuple(path[0]) == tuple(start) and tuple(path[-1]) == tuple(end)
    return d, path


@pytest.mark.parametrize("k", [2, 4, 8])
def test_fast_mode_gives_a_valid_plan_in_fewer_launches(vhp, oracle, k):
    occ = maps.maze_6()
    ny = occ.shape[0]
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    c = _ctx(vhp, occ)
    r = c.planner_solve_speculative(start, end, 0.1, 400, k=k, mode=1)
    d, path = _check_valid_plan(vhp, occ, r, start, end, 0.1)
    exact = oracle.solve(occ, start, end, 0.1, 250)
    de, _ = oracle.reconstruct_path(exact["came_from"], exact["pivots"], end)
    assert r["hits"] == 0 and r["sweeps"] < exact["n_pivots"]      # fewer sweep launches than the reference has iterations
    assert d < 1.5 * de                                              # and not a wild detour (the exact path: 1529.55)
    # labels of a lit cell: the cell is visible from its parent pivot in that pivot's own field
    py, px = np.argwhere((r["came_from"] != vhp.UNLABELLED))[::997].T
    for x, y in list(zip(px.tolist(), py.tolist()))[:12]:
        lab = int(r["came_from"][y, x])
        f = oracle.sweep_full(occ, int(r["pivots"][lab][0]), int(r["pivots"][lab][1]))
        assert f[y, x] >= 0.1


def _union_of_the_pivots_fields(oracle, occ, r, thr, start):
    """What mode 1 must leave behind, from the oracle's sweeps of the pivots it committed: the max-union, the label of the first
    pivot (in commit order) that lights a cell, and the last committed pivot's own field."""
    n = r["n_pivots"]
    union = np.zeros(occ.shape, np.float64)
    label = np.full(occ.shape, -1, np.int64)
    label[start[1], start[0]] = 0
    f = None
    for j in range(n):
        f = oracle.sweep_full(occ, int(r["pivots"][j][0]), int(r["pivots"][j][1]))
        np.maximum(union, f, out=union)
        label[(label < 0) & (f >= thr)] = j
    return union, label, f


@pytest.mark.parametrize("k", [2, 4, 8])
def test_fast_mode_fields_are_its_pivots_own(vhp, oracle, k):
    """Mode 1's slots take turns and are put back to +0.0 by the epilogue that read them, and its sweeps store nothing for dead strips:
    one stale cell would show up in the union, the labels or the local field.  The cache is left dirty by an exact-mode solve
    first, and the fast solve runs twice."""
    occ = maps.maze_6()
    ny = occ.shape[0]
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    c = _ctx(vhp, occ)
    c.planner_solve_speculative(start, end, 0.1, 250, k=8, mode=0)
    for rep in range(2):
        r = c.planner_solve_speculative(start, end, 0.1, 400, k=k, mode=1)
        assert r["status"] == 0
        union, label, last = _union_of_the_pivots_fields(oracle, occ, r, 0.1, start)
        assert np.array_equal(r["vis_global"], union), "repetition %d" % rep
        assert np.array_equal(r["vis_local"], last), "repetition %d" % rep
        got = np.where(r["came_from"] == vhp.UNLABELLED, -1, r["came_from"].astype(np.int64))
        assert np.array_equal(got, label), "repetition %d" % rep


def test_fast_mode_fields_on_a_random_map_and_an_odd_width(vhp, oracle):
    checked = 0
    for (nx, ny, seed, k) in ((301, 200, 3, 4), (256, 300, 5, 2)):
        occ = maps.random_rect_map(nx, ny, 25, 4, 40, 4, 40, seed=seed)
        free = np.argwhere(occ == 1)
        start = (int(free[0][1]), int(free[0][0]))
        end = (int(free[-1][1]), int(free[-1][0]))
        c = _ctx(vhp, occ)
        r = c.planner_solve_speculative(start, end, 0.3, 200, k=k, mode=1)
        if r["status"] not in (0, vhp.VHP_ERR_MAX_ITER):
            continue
        checked += 1
        union, label, last = _union_of_the_pivots_fields(oracle, occ, r, 0.3, start)
        assert np.array_equal(r["vis_global"], union)
        assert np.array_equal(r["vis_local"], last)
        got = np.where(r["came_from"] == vhp.UNLABELLED, -1, r["came_from"].astype(np.int64))
        assert np.array_equal(got, label)
    assert checked >= 1


def test_speculative_launches_take_the_latency_sweep(vhp):
    occ = maps.maze_6()
    ny = occ.shape[0]
    c = _ctx(vhp, occ)
    c.planner_solve_speculative((345, ny - 1 - 391), (341, ny - 1 - 10), 0.1, 250, k=4, mode=0, outputs=False)
    assert c.last_sweep_kernel() == 4


def test_edge_cases_of_the_loop_end_plain_and_exact(vhp, oracle):
    """Thresholds below zero (the loop never runs), at and above one, iteration caps of 0 and 1, start = end: the plain loop and the exact
    mode end as the oracle's solve() does, with its outputs; the fast mode must at least return."""
    occ = maps.random_rect_map(120, 90, 12, 3, 20, 3, 20, seed=4)
    free = np.argwhere(occ == 1)
    a, b = free[5], free[-7]
    start, end = (int(a[1]), int(a[0])), (int(b[1]), int(b[0]))
    c = _ctx(vhp, occ)
    for thr in (-1.0, 0.0, 0.3, 1.0, 1.5):
        for mi in (0, 1, 3, 500):
            for s, e in ((start, end), (start, start), (end, start)):
                ref = oracle.solve(occ, s, e, thr, mi)
                runs = [("plain", c.planner_solve(s, e, thr, mi))] + [("exact k=%d" % k, c.planner_solve_speculative(s, e, thr, mi, k=k, mode=0)) for k in (1, 4)]
                for what, got in runs:
                    assert got["status"] == ref["status"], (what, thr, mi, s, e)
                    if ref["status"] in (0, vhp.VHP_ERR_MAX_ITER):
                        assert np.array_equal(got["pivots"][: ref["n_pivots"] + 1], ref["pivots"]), (what, thr, mi, s, e)
                        for key in ("came_from", "vis_global", "vis_local"):
                            assert np.array_equal(got[key], ref[key]), (what, key, thr, mi, s, e)
                c.planner_solve_speculative(s, e, thr, mi, k=4, mode=1)
