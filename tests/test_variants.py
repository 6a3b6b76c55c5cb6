"""The MATLAB-flavoured variants: decay `alpha`, curve factor `fac`, the proper diagonal rule and the
min-max-scaled heuristic.

Parity chain: MATLAB itself is not available, so nothing here is pinned against it (stated in include/vhp.h, DESIGN.md and
oracle/vhp_oracle_matlab.cpp).  What is checked:
  * not gpu: the C++ restatement (oracle/vhp_oracle_matlab.cpp) against a second, independent transcription of
    getAccessibilityMap.m written here in 1-based pure Python, small grids;
  * gpu: the HIP variant kernels through the C ABI against the C++ restatement, bit for bit.
"""
import numpy as np
import pytest

import maps


def matlab_sweep_py(obst, lp, alpha, fac):
    """getAccessibilityMap.m transcribed with MATLAB's own 1-based indices (a[x][y], x = first index).  obst[x][y] = 1 free."""
    nx, ny = len(obst) - 1, len(obst[1]) - 1
    a = [[1.0] * (ny + 1) for _ in range(nx + 1)]
    L1, L2 = lp
    def run(irange, jrange, dx, dy):
        for i in irange:
            cx = L1 + dx * i
            for j in jrange:
                cy = L2 + dy * j
                if i == 0 and j == 0:
                    a[cx][cy] = 1.0
                elif i == 0:
                    a[cx][cy] = alpha * a[cx][cy - dy]
                elif j == 0:
                    a[cx][cy] = alpha * a[cx - dx][cy]
                elif i == j * fac:
                    a[cx][cy] = alpha * a[cx - dx][cy - dy]
                elif i > j * fac:
                    c = ((cy - L2) * fac) / (cx - L1)
                    p, d = a[cx - dx][cy], a[cx - dx][cy - dy]
                    f = p - c * (p - d) if dx * dy > 0 else p + c * (p - d)   # the .m files write "+ c" where c < 0
                    a[cx][cy] = alpha * f
                else:
                    c = (cx - L1) / ((cy - L2) * fac)
                    p, d = a[cx][cy - dy], a[cx - dx][cy - dy]
                    f = p - c * (p - d) if dx * dy > 0 else p + c * (p - d)
                    a[cx][cy] = alpha * f
                a[cx][cy] = a[cx][cy] * obst[cx][cy]
    run(range(0, nx - L1 + 1), range(0, ny - L2 + 1), +1, +1)   # %% 1
    run(range(0, L1), range(0, ny - L2 + 1), -1, +1)            # %% 2
    run(range(0, L1), range(0, L2), -1, -1)                     # %% 3
    run(range(0, nx - L1 + 1), range(0, L2), +1, -1)            # %% 4
    return a


@pytest.mark.parametrize("alpha,fac", [(1.0, 1.0), (0.98, 1.0), (1.0, 2.0), (0.995, 0.5), (1.0, 3.0)])
def test_oracle_restatement_equals_python_transcription(oracle, alpha, fac):
    occ = maps.random_rect_map(24, 19, 5, 2, 5, 2, 5, 3)
    ny, nx = occ.shape
    obst = [[0.0] * (ny + 1)] + [[0.0] + [float(occ[y, x]) for y in range(ny)] for x in range(nx)]
    for sx, sy in [(6, 7), (0, 0), (23, 18), (23, 0), (11, 18)]:
        if not occ[sy, sx]:
            continue
        want = matlab_sweep_py(obst, (sx + 1, sy + 1), alpha, fac)
        got = oracle.sweep_matlab(occ, sx, sy, alpha, fac)
        for x in range(nx):
            for y in range(ny):
                assert got[y, x] == want[x + 1][y + 1], "alpha %g fac %g source (%d,%d) cell (%d,%d): %r vs %r" % (
                    alpha, fac, sx, sy, x, y, got[y, x], want[x + 1][y + 1])


def test_variant_differs_from_cpp_only_where_the_algorithms_differ(oracle):
    # alpha = fac = 1 on an open grid: both algorithms give 1 everywhere they sweep; the C++ program leaves row 0 / column 0
    # unswept, MATLAB sweeps them
    occ = np.ones((40, 56), np.uint8)
    a, b = oracle.sweep_matlab(occ, 20, 13), oracle.sweep_full(occ, 20, 13)
    assert (a == 1.0).all() and (b[1:, 1:] == 1.0).all() and not b[0].any() and not b[:, 0].any()


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


@pytest.mark.gpu
@pytest.mark.parametrize("alpha,fac", [(1.0, 1.0), (0.98, 1.0), (1.0, 2.0), (0.995, 0.5)])
@pytest.mark.parametrize("nx,ny", [(104, 77), (101, 101), (1000, 1000)])
def test_variant_sweep_gpu_bit_exact(vhp, oracle, alpha, fac, nx, ny):
    occ = maps.c1_rnd1_mask() if (nx, ny) == (101, 101) else maps.random_rect_map(nx, ny, 25, 2, nx // 6, 2, ny // 6, nx + 1)
    src = np.concatenate([maps.free_sources(occ, 3, 5), np.array([(0, 0), (nx - 1, ny - 1), (nx - 1, 0)], np.int32)])
    src = src[[bool(occ[y, x]) for x, y in src]]
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.sweep_batch_variant(src, alpha, fac)
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_matlab(occ, int(sx), int(sy), alpha, fac)
        assert got[k].tobytes() == want.tobytes(), "%dx%d alpha %g fac %g source (%d,%d): %d cells differ" % (
            nx, ny, alpha, fac, sx, sy, int((got[k] != want).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("thr,alpha", [(0.2, 1.0), (0.5, 1.0), (0.2, 0.999)])
def test_variant_planner_gpu_matches_restatement(vhp, oracle, thr, alpha):
    # the exploration loop of c_sample_planner_solving_random_environments.m on BASELINE config 1's mask, (5,5) -> (95,95)
    occ = maps.c1_rnd1_mask()
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.planner_solve_variant((5, 5), (95, 95), thr, alpha, 60)
    want = oracle.solve_matlab(occ, (5, 5), (95, 95), thr, alpha, 60)
    assert got["status"] == want["status"]
    assert got["waypoints"].tolist() == want["waypoints"].tolist()
    assert got["map_builder"].tobytes() == want["map_builder"].tobytes()
    assert got["local"].tobytes() == want["local"].tobytes()
    assert np.array_equal(got["label"], want["label"])
