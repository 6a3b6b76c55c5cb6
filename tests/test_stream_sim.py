"""The streaming sweep kernel (csrc/vhp_stream.hpp) on the CPU simulator against the oracle, bit for bit.

The simulator (tests/sim) compiles the kernel's own source with 64 explicit lanes per wavefront and runs a workgroup
slot by slot, the wavefronts of a slot in a chosen order.  Results must not depend on that order (data crosses
wavefronts only across slot barriers) and must equal the oracle's field in every cell; the output starts as NaN, so
a cell no wavefront stores fails too.  No GPU needed: this is what keeps the kernel's schedule, index arithmetic and
predicates honest between GPU runs.  The GPU build of the same source is checked in tests/test_gpu_stream.py.
"""
import numpy as np
import pytest

import maps
import sim_lib


def _check(oracle, occ, src, W, order, dtype, what):
    got, st = sim_lib.sweep(occ, src, dtype, W, order)
    assert st["violations"] == 0, "%s: schedule violations %r" % (what, st)
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy)).astype(dtype)
        if got[k].tobytes() != want.tobytes():
            bad = np.argwhere(~((got[k] == want) | (np.isnan(got[k]) & np.isnan(want))))
            y, x = bad[0]
            raise AssertionError("%s, source (%d,%d): %d cells differ, first at (x=%d,y=%d): got %r want %r" % (
                what, sx, sy, len(bad), x, y, got[k][y, x], want[y, x]))
    return st


def _sources(occ, n, seed):
    ny, nx = occ.shape
    src = list(map(tuple, maps.free_sources(occ, n, seed)))
    src += [(0, 0), (nx - 1, ny - 1), (nx - 1, 0), (0, ny - 1), (min(1, nx - 1), max(ny - 2, 0)), (nx // 2, 0), (0, ny // 2)]
    src = np.array(sorted(set(src)), np.int32)
    occ[src[:, 1], src[:, 0]] = 1
    return src


# row pitch an even multiple of 64 B (all rows line-aligned), an odd one (rows alternate), thin grids, one strip, many
SIZES = [(8, 8), (8, 1), (16, 3), (8, 200), (264, 9), (104, 77), (96, 96), (200, 163), (328, 300), (640, 603), (72, 1100), (1104, 72)]


@pytest.mark.parametrize("nx,ny", SIZES)
def test_sim_small_and_ragged_grids(oracle, nx, ny):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 6, nx + ny)
    L, T, Y = sim_lib.LAZY_FLUSH, sim_lib.TWO_SLOTS, sim_lib.SMALL_Y_TEAM
    for W, order, dtype in [(4, 0 | Y, np.float64), (4, 1 | L, np.float64), (4, 2 | L | T | Y, np.float32), (8, 3, np.float64), (3, 4 | T, np.float64),
                            (4, 3 | L | Y, np.float64), (4, 3 | L | sim_lib.tile_slots(4), np.float64), (4, 2 | L | sim_lib.tile_slots(6), np.float64),
                            (4, 4 | L | sim_lib.tile_slots(8), np.float32)]:
        _check(oracle, occ, src, W, order, dtype, "%dx%d W=%d order=%d %s" % (nx, ny, W, order, dtype.__name__))


@pytest.mark.parametrize("nx,ny,W", [(1000, 1000, 4), (1024, 700, 4), (1016, 520, 8), (2048, 1500, 8), (2176, 2200, 4)])
def test_sim_multi_round_grids(oracle, nx, ny, W):
    # several rounds of W strips per octant: the round-to-round boundary rows and the diagonal hand-over at full size
    occ = maps.random_rect_map(nx, ny, 40, 5, nx // 8, 5, ny // 8, nx * 3 + ny)
    src = _sources(occ, 2, ny)[:6]
    # greedy (a wavefront runs ahead for as long as it may) and bursty interleavings: these are what an unguarded
    # buffer reuse fails under
    # (2176 x 2200: the diagonal array of the y-major units wraps, kDiagRing entries)
    _check(oracle, occ, src, W, 3 | sim_lib.LAZY_FLUSH, np.float64, "%dx%d W=%d greedy, late flushers" % (nx, ny, W))
    _check(oracle, occ, src[:3], W, 3 | sim_lib.SMALL_Y_TEAM, np.float64, "%dx%d W=%d greedy, small y-major teams" % (nx, ny, W))
    _check(oracle, occ, src[:3], W, 4 | sim_lib.TWO_SLOTS, np.float64, "%dx%d W=%d bursts, two tile slots" % (nx, ny, W))
    _check(oracle, occ, src[:3], 3, 3, np.float64, "%dx%d W=3 greedy" % (nx, ny))


def test_sim_config3_sources(oracle):
    # BASELINE config 3: the first sources of the bench batch, both wavefront orders
    occ, src = maps.config_c3(256)
    st = _check(oracle, occ, src[:6], 4, 0, np.float64, "C3")
    _check(oracle, occ, src[:6], 4, 1, np.float64, "C3 backward")
    # store shape: every store instruction is 16 bytes per lane, and the batch needs at most 1.25x the minimum number
    assert st["st8"] == 0
    assert st["st16"] * 1024 <= 1.25 * 6 * 8 * 1000 * 1000


def test_sim_open_grid_and_walls(oracle):
    occ = np.ones((136, 120), np.uint8)
    src = np.array([(60, 67), (0, 0), (119, 135), (119, 0), (0, 135)], np.int32)
    _check(oracle, occ, src, 4, 2, np.float64, "open")
    occ[40:100, 64] = 0   # a wall exactly on a 64-cell block boundary
    occ[63, 10:90] = 0
    occ[64, 30:50] = 0
    _check(oracle, occ, src, 4, 2, np.float64, "walls on block boundaries")
