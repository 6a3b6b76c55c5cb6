"""The octant/front decomposition the HIP kernels use is bit-identical to the oracle."""
import subprocess
import os

import numpy as np
import pytest

import maps
from schedule_model import sweep_units


def _check(oracle, occ, sx, sy):
    want = oracle.sweep_full(occ, sx, sy)
    got = sweep_units(occ, sx, sy)
    assert not np.isnan(got).any(), "a cell was left unwritten"
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_model_random_maps(oracle, seed):
    occ = maps.random_rect_map(101, 77, 25, 2, 20, 2, 20, seed)
    src = maps.free_sources(occ, 12, seed + 100)
    for sx, sy in src:
        _check(oracle, occ, int(sx), int(sy))


def test_model_border_sources(oracle):
    occ = maps.random_rect_map(40, 33, 10, 2, 8, 2, 8, 5)
    occ[0, :] = 1
    occ[:, 0] = 1
    occ[-1, :] = 1
    occ[:, -1] = 1
    for sx, sy in [(0, 0), (39, 0), (0, 32), (39, 32), (0, 17), (20, 0), (39, 5), (7, 32), (1, 1)]:
        _check(oracle, occ, sx, sy)


def test_model_blocked_source(oracle):
    occ = np.ones((20, 20), np.uint8)
    occ[5, 6] = 0
    _check(oracle, occ, 6, 5)


def test_model_thin_grids(oracle):
    for nx, ny in [(1, 1), (1, 9), (9, 1), (2, 2), (3, 50), (50, 3)]:
        occ = np.ones((ny, nx), np.uint8)
        for sx in range(nx):
            for sy in {0, ny // 2, ny - 1}:
                _check(oracle, occ, sx, sy)


def test_markstein_division_exact():
    """c = j/i via (table reciprocal, mul, 2 fma) is bit-exact for all j < i <= 8192."""
    exe = os.path.join(maps.GOLDEN, "..", "..", "oracle", "markstein_check")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe), "markstein_check"])
    subprocess.check_call([exe, "8192"], stdout=subprocess.DEVNULL)


def test_argmin_rank_model(oracle):
    """numpy statement of the planner step: h per lit cell + lexicographic (h, push rank) min
    reproduces the oracle's heap top(), including ties."""
    from schedule_model import first_touch_rank
    from oracle_lib import UNLABELLED
    for seed in (1, 2, 3, 4):
        occ = maps.random_rect_map(61, 53, 8, 3, 12, 3, 12, seed)
        ny, nx = occ.shape
        pts = maps.free_sources(occ, 2, seed)
        (sx, sy), (ex, ey) = [tuple(int(v) for v in q) for q in pts]
        thr = 0.3
        came = np.full((ny, nx), UNLABELLED, np.uint64)
        came[sy, sx] = 0
        vg = np.zeros((ny, nx))
        piv = np.array([[sx, sy]], np.int32)
        step = oracle.planner_step(occ, (sx, sy), (ex, ey), thr, 0, piv, came, vg)
        # model: after the sweep every lit cell has h = scale*vg + (d(cell,end) + d(cell,parent))
        scale = np.sqrt(float(ny * ny + nx * nx))
        ys, xs = np.nonzero(vg >= thr)
        xs, ys = xs[(xs > 0) | (sx == 0)], ys[(xs > 0) | (sx == 0)]  # never-swept column 0 / row 0 are not pushed
        keep = (ys > 0) | (sy == 0)
        xs, ys = xs[keep], ys[keep]
        d_end = np.sqrt(((xs - ex) ** 2 + (ys - ey) ** 2).astype(np.float64))
        d_par = np.sqrt(((xs - sx) ** 2 + (ys - sy) ** 2).astype(np.float64))
        h = scale * vg[ys, xs] + (d_end + d_par)
        rank = np.array([first_touch_rank(nx, ny, sx, sy, int(x), int(y)) for x, y in zip(xs, ys)], np.int64)
        order = np.lexsort((rank, h))
        assert (int(xs[order[0]]), int(ys[order[0]])) == step["top"]
        assert h[order[0]] == step["top_h"]
