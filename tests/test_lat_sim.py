"""The latency sweep (csrc/vhp_lat.hpp) on the CPU simulator against the oracle, bit for bit.

Same unit code as the pool sweep, another schedule: one workgroup per octant, strip p bound to wavefront p mod W (the last
wavefront of a y-major workgroup runs the diagonal).  The simulator (tests/sim/vhp_pool_sim.cpp) runs every wavefront as a
coroutine under several scheduling policies; the fields must equal the oracle's in every cell (output, LDS and scratch
start as NaN or as entries of "an earlier launch"), and no run may end with every wavefront waiting -- including the runs with
fewer wavefronts than strips, where a wavefront sweeps several strips one after the other.  No GPU needed; the gfx950 build of
the same source is checked in tests/test_gpu_lat.py.
"""
import numpy as np
import pytest

import maps
import sim_lib
from sim_lib import POOL_BACKWARD, POOL_BURSTS, POOL_GREEDY, POOL_POINTS_ALWAYS, POOL_POINTS_RANDOM, POOL_RANDOM, POOL_ROUND_ROBIN


def _check(oracle, occ, src, what, dtype=np.float64, **kw):
    got, st = sim_lib.lat_sweep(occ, src, dtype, **kw)
    assert st["deadlock"] == 0, "%s: every wavefront waiting %r" % (what, st)
    assert st["err"] == 0
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


SIZES = [(8, 8), (8, 1), (16, 3), (8, 200), (264, 9), (104, 77), (96, 96), (200, 163), (328, 300), (640, 603), (72, 1100), (1104, 72),
         (2, 5), (10, 9), (106, 77), (130, 131), (690, 402),  # (the planner's maze is 690 wide)
         (1, 1), (3, 7), (9, 9), (101, 101), (105, 78), (263, 300), (689, 402), (71, 1100), (971, 65)]  # odd widths: rows off the 16-byte grid (C1 is 101 x 101)
SHAPES = [  # W wavefronts per workgroup, policy, dtype
    (12, POOL_ROUND_ROBIN | POOL_POINTS_ALWAYS, np.float64),
    (12, POOL_RANDOM | POOL_POINTS_RANDOM, np.float32),
    (4, POOL_GREEDY | POOL_POINTS_ALWAYS, np.float64),     # fewer wavefronts than strips on the larger grids: rounds
    (2, POOL_BACKWARD | POOL_POINTS_ALWAYS, np.float64),   # one strip wavefront + the diagonal's in a y-major workgroup
    (3, POOL_BURSTS | POOL_POINTS_RANDOM, np.float64),
]


@pytest.mark.parametrize("nx,ny", SIZES)
def test_lat_sim_small_and_ragged_grids(oracle, nx, ny):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 4, nx + ny)
    for W, policy, dtype in SHAPES:
        _check(oracle, occ, src, "%dx%d W=%d policy=%d" % (nx, ny, W, policy), dtype, W=W, policy=policy, seed=nx + W)


def test_lat_sim_config2_shape(oracle):
    """The C2 launch itself: 1000 x 1000, empty, one source in the centre; every hand-off of the static schedule comes out of the
    writer's ring when the wavefronts run in lockstep."""
    occ = np.ones((1000, 1000), np.uint8)
    st = _check(oracle, occ, np.array([[500, 500]], np.int32), "C2", W=12, policy=POOL_ROUND_ROBIN)
    assert st["from_ring"] > 0


def test_lat_sim_config3_map_three_sources(oracle):
    occ = maps.random_rect_map(1000, 1000, 50, 20, 100, 20, 100, 1)
    src = maps.free_sources(occ, 3, 11)
    _check(oracle, occ, src, "C3 map", W=12, policy=POOL_RANDOM | POOL_POINTS_RANDOM, seed=5)
    _check(oracle, occ, src, "C3 map, rounds", W=5, policy=POOL_GREEDY, seed=6)


def _walled(nx, ny, seed, density):
    """A map where light dies early: random cells blocked with the given density, plus a few full walls."""
    rng = np.random.RandomState(seed)
    occ = (rng.rand(ny, nx) >= density).astype(np.uint8)
    for k in range(3):
        occ[rng.randint(0, ny), :] = 0
        occ[:, rng.randint(0, nx)] = 0
    return occ


@pytest.mark.parametrize("nx,ny,density", [(200, 163, 0.5), (328, 300, 0.2), (640, 603, 0.08), (1104, 72, 0.3), (72, 1100, 0.3), (130, 131, 0.95), (201, 163, 0.5), (329, 301, 0.2), (131, 130, 0.95)])
def test_lat_sim_strips_that_die(oracle, nx, ny, density):
    """Maps where every strip's values turn +0.0 long before its march ends (a strip that is dead, below a dead strip, stops sweeping
    and stores zeros): the fields must still equal the oracle's bit for bit, whatever the order the wavefronts run in."""
    occ = _walled(nx, ny, nx + ny, density)
    src = _sources(occ, 5, nx * 3 + ny)
    for W, policy, dtype in SHAPES:
        _check(oracle, occ, src, "%dx%d density %.2f W=%d policy=%d" % (nx, ny, density, W, policy), dtype, W=W, policy=policy, seed=ny + W)


def test_lat_sim_source_in_a_box(oracle):
    # the source walled in: everything outside the box is dark, and so is every strip that starts outside it
    occ = np.ones((300, 328), np.uint8)
    occ[100:141, 150] = 0; occ[100:141, 190] = 0; occ[100, 150:191] = 0; occ[140, 150:191] = 0
    src = np.array([[170, 120], [151, 101], [189, 139], [10, 10]], np.int32)
    st = _check(oracle, occ, src, "box", W=8, policy=POOL_RANDOM | POOL_POINTS_RANDOM, seed=3)
    assert st["died"] > 0, st
    _check(oracle, occ, src, "box, rounds", W=3, policy=POOL_BACKWARD | POOL_POINTS_ALWAYS, seed=4)


def test_lat_sim_maze_6_pivots(oracle):
    # the planner's map (690 x 402) from a few of the pivots of its reference run: the light is gone after ~60 steps of marches of ~550
    occ = maps.maze_6()
    src = np.array([[345, 391], [341, 10], [112, 201], [600, 60], [689, 401], [0, 0]], np.int32)
    src = src[[bool(occ[y, x]) for x, y in src]]
    st = _check(oracle, occ, src, "maze_6", W=8, policy=POOL_ROUND_ROBIN | POOL_POINTS_ALWAYS, seed=1)
    assert st["died"] > 20 * len(src), st   # (most of the ~60 strips of a source die)
    _check(oracle, occ, src, "maze_6", W=8, policy=POOL_BURSTS | POOL_POINTS_RANDOM, seed=2)


@pytest.mark.parametrize("nx,ny", [(16, 375), (8, 571), (375, 16), (9, 300), (40, 333)])
def test_lat_sim_thin_salted_grids(oracle, nx, ny):
    """Thin grids with half the cells blocked at random, sources anywhere: light that survives only in the sub-diagonal cell next to a
    blocked diagonal cell (the stale diagonal, SURVEY Q1: diag(t) = V(t, t - 1) * occ(t, t)) must not be taken for a dead band -- a
    fuzzing run found exactly that on a 16 x 375 map while band 0 kept the sub-diagonal in a register of its own."""
    for seed in range(6):
        rng = np.random.RandomState(1000 * nx + ny + seed)
        occ = (rng.rand(ny, nx) >= 0.5).astype(np.uint8)
        src = np.stack([rng.randint(0, nx, 3), rng.randint(0, ny, 3)], 1).astype(np.int32)
        occ[src[:, 1], src[:, 0]] = 1
        for W, policy, dtype in SHAPES[:3]:
            _check(oracle, occ, src, "%dx%d salt seed %d W=%d" % (nx, ny, seed, W), dtype, W=W, policy=policy, seed=seed)


@pytest.mark.parametrize("nx,ny,density", [(328, 300, 0.0), (200, 520, 0.02), (640, 603, 0.08), (1104, 72, 0.3), (72, 1100, 0.3), (329, 301, 0.2)])
def test_lat_sim_two_workgroups_per_unit(oracle, nx, ny, density):
    """The long octants' launch shape (LatArgs::halves = 2, 4 or 8): the bands of an octant dealt out to that many workgroups, band p to workgroup
    (p / W) % halves -- a band whose band below lives in the other workgroup reads that band's line in global memory, block by block, and its
    death out of a record beside the lines.  With one sweeper per workgroup EVERY band reads across; with two or three, every second or
    third; bands that die before, while and after their reader starts (walls of every density), every order of the wavefronts."""
    occ = _walled(nx, ny, nx + 2 * ny, density) if density > 0 else np.ones((ny, nx), np.uint8)
    src = _sources(occ, 3, nx + ny)
    for W, policy, halves in [(1, POOL_ROUND_ROBIN, 2), (1, POOL_BACKWARD | POOL_POINTS_ALWAYS, 2), (2, POOL_RANDOM | POOL_POINTS_RANDOM, 2), (3, POOL_BURSTS, 2),
                              (8, POOL_GREEDY | POOL_POINTS_RANDOM, 2), (1, POOL_RANDOM | POOL_POINTS_ALWAYS, 4), (2, POOL_BURSTS | POOL_POINTS_RANDOM, 4), (1, POOL_GREEDY, 8)]:
        dtype = np.float32 if (W == 3) else np.float64
        _check(oracle, occ, src, "%dx%d density %.2f W=%d policy=%d, %d workgroups per unit" % (nx, ny, density, W, policy, halves), dtype, W=W, policy=policy, seed=nx + W,
               halves=halves)
