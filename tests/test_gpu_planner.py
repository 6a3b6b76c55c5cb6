"""Parity of the GPU planner (vhp_planner_solve through the C ABI) with the CPU oracle:
pivots, labels (cameFrom / lightSource_enum), global and local visibility -- all bit-exact."""
import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def _solve_both(vhp, oracle, occ, start, end, thr, max_iter):
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.planner_solve(start, end, thr, max_iter)
    want = oracle.solve(occ, start, end, thr, max_iter)
    return c, got, want


def _assert_same_solution(got, want, what):
    assert got["status"] == want["status"], "%s: status %d vs %d" % (what, got["status"], want["status"])
    assert got["n_pivots"] == want["n_pivots"], "%s: %d pivots vs %d" % (what, got["n_pivots"], want["n_pivots"])
    assert got["pivots"].tolist() == want["pivots"].tolist(), what + ": pivot list differs"
    for name in ("came_from", "vis_global", "vis_local"):
        if got[name].tobytes() != want[name].tobytes():
            bad = np.argwhere(got[name] != want[name])
            y, x = bad[0]
            raise AssertionError("%s: %s differs in %d cells, first (x=%d,y=%d): %r vs %r" % (
                what, name, len(bad), x, y, got[name][y, x], want[name][y, x]))


@pytest.mark.parametrize("seed,thr", [(1, 0.25), (2, 0.5), (3, 0.1), (4, 0.25), (5, 0.9)])
def test_random_maps(vhp, oracle, seed, thr):
    occ = maps.random_rect_map(160, 131, 22, 4, 30, 4, 30, seed)
    pts = maps.free_sources(occ, 2, seed + 50)
    start, end = tuple(int(v) for v in pts[0]), tuple(int(v) for v in pts[1])
    _, got, want = _solve_both(vhp, oracle, occ, start, end, thr, 60)
    _assert_same_solution(got, want, "seed %d thr %g" % (seed, thr))


def test_config4_maze6(vhp, oracle):
    # BASELINE config 4: maze_6, thr 0.1, start {345,391}, end {341,10} (mode 2 => y flipped)
    occ = maps.maze_6()
    ny = occ.shape[0]
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    c, got, want = _solve_both(vhp, oracle, occ, start, end, 0.1, 250)
    _assert_same_solution(got, want, "maze_6")
    assert got["n_pivots"] == 64
    d, path = c.reconstruct_path(got["came_from"], got["pivots"], end)
    dw, pathw = oracle.reconstruct_path(want["came_from"], want["pivots"], end)
    assert d == dw and path.tolist() == pathw.tolist()
    assert "%.6g" % d == "1529.55"


def test_config1_rnd1_mask_planner(vhp, oracle):
    # BASELINE config 1's grid (tests/golden/c1_rnd1_mask.npz, from MATLAB_code/rnd_1.mat), start (5,5) -> end (95,95)
    # as in f_comparison_to_a_star.m:30-31, at the paper's threshold and at the script's
    occ = maps.c1_rnd1_mask()
    for thr in (0.5, 0.2):
        c, got, want = _solve_both(vhp, oracle, occ, (5, 5), (95, 95), thr, 120)
        assert c.last_sweep_kernel() == 4  # the latency sweep runs the loop on this odd width too (round 4)
        _assert_same_solution(got, want, "C1 rnd_1 mask thr %g" % thr)
        if got["status"] == vhp.VHP_OK:
            d, path = c.reconstruct_path(got["came_from"], got["pivots"], (95, 95))
            dw, pathw = oracle.reconstruct_path(want["came_from"], want["pivots"], (95, 95))
            assert d == dw and path.tolist() == pathw.tolist()


def test_negative_threshold_skips_the_loop(vhp, oracle):
    # while (visibility_global_(end) <= threshold) with an all-zero field: a negative threshold never enters the
    # loop (solver.cpp:127), lightSources_[0] becomes `end` (:141), nothing is swept
    occ = maps.random_rect_map(64, 48, 6, 3, 9, 3, 9, 3)
    pts = maps.free_sources(occ, 2, 1)
    start, end = tuple(int(v) for v in pts[0]), tuple(int(v) for v in pts[1])
    _, got, want = _solve_both(vhp, oracle, occ, start, end, -0.5, 10)
    _assert_same_solution(got, want, "negative threshold")
    assert got["n_pivots"] == 0 and got["pivots"].tolist() == [list(end)] and not got["vis_global"].any()


def test_device_resident_results(vhp, oracle):
    # vhp_planner_solve_device + vhp_planner_results_device: labels, union, local field and pivots read back from the
    # device arrays equal the oracle's (and the host-copy entry point's)
    import ctypes as C
    import torch
    occ = maps.maze_6()
    ny, nx = occ.shape
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    c = vhp.Context(0)
    c.set_map(occ)
    rc, n_piv, ptr = c.planner_solve_device(start, end, 0.1, 250)
    want = oracle.solve(occ, start, end, 0.1, 250)
    assert rc == want["status"] == 0 and n_piv == want["n_pivots"] == 64
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def fetch(p, shape, dtype):
        a = np.empty(shape, dtype)
        assert hip.hipMemcpy(a.ctypes.data, p, a.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return a

    lab = fetch(ptr["labels"], (ny, nx), np.uint32).astype(np.uint64)
    lab[lab == 0xFFFFFFFF] = vhp.UNLABELLED
    assert np.array_equal(lab, want["came_from"])
    assert fetch(ptr["vis_global"], (ny, nx), np.float64).tobytes() == want["vis_global"].tobytes()
    assert fetch(ptr["vis_local"], (ny, nx), np.float64).tobytes() == want["vis_local"].tobytes()
    assert fetch(ptr["pivots"], (n_piv + 1, 2), np.int32).tolist() == want["pivots"].tolist()


def test_1000_shipped_config_seed1(vhp, oracle):
    # the shipped settings.config with seedValue 1: pivots (50,50),(273,350),(525,675), path 1346.71
    import platform
    if platform.libc_ver()[1] != "2.35":
        pytest.skip("glibc rand() stream differs")
    occ = oracle.generate_env(1000, 1000, 15, 100, 200, 100, 200, 1)
    c, got, want = _solve_both(vhp, oracle, occ, (50, 50), (990, 990), 0.25, 250)
    _assert_same_solution(got, want, "1000^2 seed 1")
    assert got["pivots"].tolist() == [[50, 50], [273, 350], [525, 675], [990, 990]]
    d, _ = c.reconstruct_path(got["came_from"], got["pivots"], (990, 990))
    assert "%.6g" % d == "1346.71"


def test_large_grid_planner(vhp, oracle):
    occ = maps.random_rect_map(2300, 2100, 30, 60, 300, 60, 300, 8)
    pts = maps.free_sources(occ, 2, 3)
    start, end = tuple(int(v) for v in pts[0]), tuple(int(v) for v in pts[1])
    _, got, want = _solve_both(vhp, oracle, occ, start, end, 0.3, 40)
    _assert_same_solution(got, want, "2300x2100")


def test_max_iter_livelock(vhp, oracle):
    # maze_6 at thr 0.25 repeats a pivot until max_iter; outputs still match
    occ = maps.maze_6()
    ny = occ.shape[0]
    _, got, want = _solve_both(vhp, oracle, occ, (345, ny - 1 - 391), (341, ny - 1 - 10), 0.25, 40)
    assert got["status"] == vhp.VHP_ERR_MAX_ITER
    _assert_same_solution(got, want, "livelock")


def test_validation_codes(vhp):
    occ = np.ones((8, 8), np.uint8)
    occ[2, 2] = 0
    c = vhp.Context(0)
    c.set_map(occ)
    assert c.planner_solve((9, 1), (1, 1), 0.5, 5)["status"] == vhp.VHP_ERR_START_OOB
    assert c.planner_solve((1, 1), (1, 8), 0.5, 5)["status"] == vhp.VHP_ERR_END_OOB
    assert c.planner_solve((2, 2), (1, 1), 0.5, 5)["status"] == vhp.VHP_ERR_START_OCCUPIED
    assert c.planner_solve((1, 1), (2, 2), 0.5, 5)["status"] == vhp.VHP_ERR_END_OCCUPIED


def test_tie_break_symmetric_map(vhp, oracle):
    # Q6: on a symmetric empty map many cells share the minimal h; the first-pushed wins
    occ = np.ones((65, 65), np.uint8)
    _, got, want = _solve_both(vhp, oracle, occ, (32, 32), (64, 64), 0.5, 10)
    _assert_same_solution(got, want, "symmetric")
    occ[20:45, 40] = 0
    occ[40, 20:45] = 0
    _, got, want = _solve_both(vhp, oracle, occ, (32, 32), (64, 64), 0.5, 20)
    _assert_same_solution(got, want, "symmetric with walls")


def test_queue_variant_small(vhp, oracle):
    # computeVisibilityUsingQueue: literal device emulation, bit-exact incl. pop-order effects (Q8)
    occ = maps.random_rect_map(120, 97, 40, 2, 9, 2, 9, 7)
    src = maps.free_sources(occ, 6, 7)
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.sweep_batch(src, variant=vhp.SWEEP_QUEUE)
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_queue(occ, int(sx), int(sy))
        assert got[k].tobytes() == want.tobytes(), "queue variant differs for source %d" % k
    # border sources: size_t wrap of ls-1 in the reference == rejected by isValid
    src = np.array([(0, 0), (119, 96), (0, 50), (60, 0)], np.int32)
    src = src[[bool(occ[y, x]) for x, y in src]]
    got = c.sweep_batch(src, variant=vhp.SWEEP_QUEUE)
    for k, (sx, sy) in enumerate(src):
        assert got[k].tobytes() == oracle.sweep_queue(occ, int(sx), int(sy)).tobytes()


def test_plain_speculative_plain_on_one_context(vhp, oracle):
    # The plain solve keeps two local fields that take turns (dark cells of a sweep stay unwritten; vhp_planner.hip.h), the
    # speculative solve one, and both share the context's planner state: alternate them, with solves of an odd and of an even
    # number of iterations, and compare everything -- the local field, wherever it ended up, included.
    occ = maps.random_rect_map(160, 132, 22, 4, 30, 4, 30, 7)
    pts = maps.free_sources(occ, 6, 57)
    c = vhp.Context(0)
    c.set_map(occ)
    seen = set()
    for k in range(5):
        start, end = tuple(int(v) for v in pts[k]), tuple(int(v) for v in pts[k + 1])
        want = oracle.solve(occ, start, end, 0.3, 80)
        got = c.planner_solve(start, end, 0.3, 80)
        _assert_same_solution(got, want, "plain solve %d" % k)
        seen.add(want["n_pivots"] & 1)
        spec = c.planner_solve_speculative(start, end, 0.3, 80, 4, 0)
        _assert_same_solution(spec, want, "speculative solve %d" % k)
    assert seen == {0, 1}, "pick sources that give solves of both parities (%r)" % seen
