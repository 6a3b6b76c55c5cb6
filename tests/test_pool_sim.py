"""The pool sweep kernel (csrc/vhp_pool.hpp) on the CPU simulator against the oracle, bit for bit.

The simulator (tests/sim/vhp_pool_sim.cpp) compiles the kernel's own source with 64 explicit lanes per wavefront and runs
every wavefront of G workgroups as a coroutine: backoff() (a wavefront that has to wait) switches to the scheduler, and so
do -- always or at random -- the marked places where a device wavefront can be overtaken between a read and the
compare-and-swap that depends on it.  Whatever the interleaving, the fields must equal the oracle's in every cell (the
output, the LDS and the diagonal scratch start as NaN), every unit must have been pulled exactly once, and no run may end
with every wavefront waiting.  No GPU needed; the gfx950 build of the same source is checked in tests/test_gpu_pool.py.
"""
import numpy as np
import pytest

import maps
import sim_lib
from sim_lib import (POOL_BACKWARD, POOL_BURSTS, POOL_GREEDY, POOL_POINTS_ALWAYS, POOL_POINTS_RANDOM, POOL_RANDOM, POOL_ROUND_ROBIN,
                     POOL_SHUFFLED_QUEUE, POOL_TIGHT_BUSY_CAP)


def _check(oracle, occ, src, what, dtype=np.float64, **kw):
    got, st = sim_lib.pool_sweep(occ, src, dtype, **kw)
    assert st["deadlock"] == 0, "%s: every wavefront waiting %r" % (what, st)
    assert st["err"] == 0
    assert st["pulled"] >= 8 * len(src), "%s: %d units pulled of %d" % (what, st["pulled"], 8 * len(src))
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
         # widths that are not a multiple of 8 (the kernel's ANYW build: 8-column windows leave as they are swept), even and odd
         (1, 1), (1, 70), (3, 2), (7, 9), (12, 9), (10, 140), (101, 101), (202, 163), (333, 77), (37, 300), (690, 402), (1101, 70)]
SHAPES = [  # W wavefronts, C contexts, G workgroups, policy, dtype
    (12, 4, 1, POOL_ROUND_ROBIN | POOL_POINTS_ALWAYS, np.float64),
    (12, 4, 2, POOL_RANDOM | POOL_POINTS_RANDOM, np.float64),
    (4, 2, 1, POOL_GREEDY | POOL_POINTS_ALWAYS, np.float32),
    (6, 4, 3, POOL_BURSTS | POOL_POINTS_RANDOM | POOL_SHUFFLED_QUEUE | POOL_TIGHT_BUSY_CAP, np.float64),
    (2, 1, 1, POOL_BACKWARD | POOL_POINTS_ALWAYS, np.float64),
    (1, 3, 1, POOL_ROUND_ROBIN, np.float64),   # one wavefront does everything: no interleaving can be needed for progress
    (16, 8, 1, POOL_GREEDY, np.float64),
]


@pytest.mark.parametrize("nx,ny", SIZES)
def test_pool_sim_small_and_ragged_grids(oracle, nx, ny):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 6, nx + ny)
    for W, C, G, policy, dtype in SHAPES:
        _check(oracle, occ, src, "%dx%d W=%d C=%d G=%d policy=%d %s" % (nx, ny, W, C, G, policy, dtype.__name__), dtype,
               W=W, C=C, G=G, policy=policy, seed=nx + ny)


@pytest.mark.parametrize("nx,ny", [(208, 77), (104, 90), (200, 163), (101, 50), (64, 64)])
@pytest.mark.parametrize("offset", [1, 2, 5, 8, 15])
def test_pool_sim_fields_off_the_line_grid(oracle, nx, ny, offset):
    # the fields begin `offset` cells into a line: the ANYW build of the kernel also on widths that are a multiple of 8 or 16 (every
    # row at the same place in its line: one set of rows, or two), pairs off the 16-byte grid for odd offsets; nothing outside the fields
    occ = maps.random_rect_map(nx, ny, 12, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 3 + ny + offset)
    src = _sources(occ, 3, nx + offset)
    for dtype in (np.float64, np.float32):
        got, st = sim_lib.pool_sweep(occ, src, dtype, W=9, C=3, G=2, policy=POOL_RANDOM | POOL_POINTS_RANDOM, seed=offset, out_offset=offset)
        assert st["deadlock"] == 0 and st["err"] == 0
        for k, (sx, sy) in enumerate(src):
            want = oracle.sweep_full(occ, int(sx), int(sy)).astype(dtype)
            assert got[k].tobytes() == want.tobytes(), "%dx%d offset %d %s source (%d,%d)" % (nx, ny, offset, dtype.__name__, sx, sy)


@pytest.mark.parametrize("nx,ny", [(1000, 1000), (1024, 700), (1016, 520), (2048, 1500), (2176, 2200), (1002, 700), (1001, 971), (2049, 1100)])
def test_pool_sim_large_grids(oracle, nx, ny):
    # full-size octants: up to 34 strips of one unit, marches of up to 35 blocks
    occ = maps.random_rect_map(nx, ny, 40, 5, nx // 8, 5, ny // 8, nx * 3 + ny)
    src = _sources(occ, 2, ny)[:5]
    _check(oracle, occ, src, "%dx%d random" % (nx, ny), W=12, C=4, G=1, policy=POOL_RANDOM | POOL_POINTS_RANDOM, seed=3)
    _check(oracle, occ, src[:3], "%dx%d greedy" % (nx, ny), W=12, C=4, G=2, policy=POOL_GREEDY | POOL_POINTS_ALWAYS, seed=4)
    _check(oracle, occ, src[:3], "%dx%d narrow pool" % (nx, ny), W=2, C=1, G=1, policy=POOL_BURSTS | POOL_POINTS_RANDOM, seed=5)


def test_pool_sim_every_hand_off_path_is_taken(oracle):
    # a strip takes the boundary values of the strip below out of that wavefront's LDS ring (fast) or, a block at a time,
    # out of global memory (the writer is far ahead, has moved on, or overwrote the entries while they were copied): each of
    # these must happen in some interleaving here, and none may change a bit
    occ = maps.random_rect_map(640, 603, 40, 1, 80, 1, 75, 640 * 7 + 603)
    src = maps.free_sources(occ, 4, 5)
    seen = dict(from_ring=0, from_global=0, too_far=0, overwritten=0)
    for W, C, G, policy in [(12, 4, 1, POOL_ROUND_ROBIN | POOL_POINTS_ALWAYS), (2, 1, 1, POOL_BACKWARD | POOL_POINTS_ALWAYS),
                            (8, 3, 1, POOL_GREEDY | POOL_POINTS_RANDOM)]:
        st = _check(oracle, occ, src, "W=%d C=%d policy=%d" % (W, C, policy), W=W, C=C, G=G, policy=policy)
        for k in seen:
            seen[k] += st[k]
    assert all(v > 0 for v in seen.values()), seen


def test_pool_sim_config3_sources(oracle):
    # BASELINE config 3: the first sources of the bench batch; every store instruction is 16 bytes per lane, and the batch
    # needs at most 1.25x the minimum number of them
    occ, src = maps.config_c3(256)
    st = _check(oracle, occ, src[:6], "C3", W=12, C=4, G=2, policy=POOL_RANDOM | POOL_POINTS_RANDOM)
    assert st["st8"] == 0
    assert st["st16"] * 1024 <= 1.25 * 6 * 8 * 1000 * 1000


def test_pool_sim_open_grid_walls_and_a_rejected_source(oracle):
    occ = np.ones((136, 120), np.uint8)
    src = np.array([(60, 67), (0, 0), (119, 135), (119, 0), (0, 135)], np.int32)
    _check(oracle, occ, src, "open", W=5, C=3, policy=POOL_RANDOM | POOL_POINTS_ALWAYS)
    occ[40:100, 64] = 0   # a wall exactly on a 64-cell block boundary
    occ[63, 10:90] = 0
    occ[64, 30:50] = 0
    _check(oracle, occ, src, "walls on block boundaries", W=5, C=3, policy=POOL_RANDOM | POOL_POINTS_ALWAYS)
    # a source outside the grid raises the error flag, and its units do nothing
    bad = np.array([(60, 67), (120, 5)], np.int32)
    got, st = sim_lib.pool_sweep(occ, bad, W=4, C=2)
    assert st["err"] == 1 and st["deadlock"] == 0
    assert got[0].tobytes() == oracle.sweep_full(occ, 60, 67).tobytes() and np.isnan(got[1]).all()


def test_pool_sim_first_units_by_workgroup_index(oracle):
    # Args::static_round: the first unit of every context by workgroup index, the queue behind them -- every unit still swept exactly
    # once, whether the launch has just enough units for the contexts (8 sources = 64 units, 4 workgroups x 4 contexts x ... ) or many
    # more, whatever the number of head contexts (seed % 3 + 1), and also when the queue is empty before a context took its first
    occ = maps.random_rect_map(200, 163, 12, 4, 30, 4, 30, 5)
    occ[:2, :2] = 1
    seen = set()
    for n_src, W, C, G, seed in [(2, 4, 4, 4, 1), (8, 6, 3, 2, 3), (24, 4, 2, 5, 7), (8, 12, 4, 16, 9), (9, 3, 3, 3, 1), (6, 2, 4, 12, 3), (1, 4, 4, 4, 1)]:
        src = _sources(occ, n_src, seed)[:n_src] if n_src > 1 else np.array([(1, 1)], np.int32)
        for policy in (POOL_ROUND_ROBIN | POOL_POINTS_ALWAYS, POOL_RANDOM | POOL_POINTS_RANDOM, POOL_BACKWARD, POOL_GREEDY | POOL_TIGHT_BUSY_CAP):
            st = _check(oracle, occ, src, "static round n=%d W=%d C=%d G=%d seed=%d policy=%d" % (n_src, W, C, G, seed, policy),
                        W=W, C=C, G=G, policy=policy, seed=seed)
            assert st["static_round"] == (1 if 8 * len(src) >= C * G else 0)
            seen.add(st["static_round"])
    assert seen == {0, 1}


_ASAN_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import maps, sim_lib
for nx, ny in [(7, 9), (12, 9), (104, 77), (101, 101), (202, 163), (333, 77), (264, 200)]:
    occ = maps.random_rect_map(nx, ny, max(3, min(40, nx * ny // 400)), 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = np.array([(0, 0), (nx - 1, ny - 1), (nx - 1, 0), (0, ny - 1), (nx // 2, ny // 2), (1, max(ny - 2, 0))], np.int32)
    for dtype in (np.float64, np.float32):
        # the fields exactly as large as they are: a store one cell outside is a store outside the allocation
        got, st = sim_lib.pool_sweep(occ, src, dtype, W=6, C=3, G=2, policy=sim_lib.POOL_RANDOM | sim_lib.POOL_POINTS_RANDOM, seed=nx)
        assert st["deadlock"] == 0 and not np.isnan(got).any()
        got, st = sim_lib.lat_sweep(occ, src, dtype, W=8, policy=2, seed=ny)
        assert st["deadlock"] == 0 and not np.isnan(got).any()
print("asan-clean")
"""


def test_sims_under_address_sanitizer(tmp_path):
    # The kernels' own source under AddressSanitizer, in a child process (the sanitizer's runtime has to be loaded first): every
    # global and LDS access of both persistent kernels inside its allocation, on widths of every residue.  GPU sanitizers are not
    # available on the test pool; this is the CPU build's.
    import os
    import shutil
    import subprocess
    import sys
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    asan = subprocess.check_output([gxx, "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "asan_child.py"
    script.write_text(_ASAN_CHILD % dict(tests=here, root=os.path.dirname(here)))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", VHP_SIM_LIB="libvhp_pool_sim_asan.so")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan-clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
