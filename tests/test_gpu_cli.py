"""End-to-end: the `vhp` command-line driver (settings.config in, output/*.txt out) produces
byte-identical files to what the reference's writers would emit for the oracle's arrays."""
import os
import subprocess

import numpy as np
import pytest

import host_lib
import maps

pytestmark = pytest.mark.gpu

BASE = """mode={mode}
ncols={nx}
nrows={ny}
nb_of_obstacles={nb}
minWidth=4
maxWidth=30
minHeight=4
maxHeight=30
randomSeed=0
seedValue={seed}
imagePath={image}
start={{{sx},{sy}}}
end={{{ex},{ey}}}
max_iter={max_iter}
visibilityThreshold={thr}
lightStrength=1
timer=1
saveResults=1
saveCameFrom=1
saveLightSources=1
saveGlobalVisibility=1
saveLocalVisibility=1
saveVisibilityField=1
silent=0
ballRadius=4
"""


def _run(tmp_path, cfg):
    if not os.path.exists(host_lib.CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(host_lib.PKG, "host")])
    (tmp_path / "config").mkdir(exist_ok=True)
    (tmp_path / "config" / "settings.config").write_text(cfg)
    r = subprocess.run([host_lib.CLI], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    return r


def _read(tmp_path, name):
    return (tmp_path / "output" / name).read_text()


YELLOW, BLACK, RED, MAGENTA, CYAN, GREEN, WHITE = ((255, 255, 0), (0, 0, 0), (255, 0, 0), (255, 0, 255), (0, 255, 255), (0, 255, 0),
                                                    (255, 255, 255))


def _disc(img, cx, cy, r, colour, ring=False):
    # the reference's two ball loops (solver.cpp:920-943): isValid() rejects coordinates outside the image
    ny, nx = img.shape[:2]
    for j in range(-r - 1, r + 2):
        for k in range(-r - 1, r + 2):
            x, y = cx + j, cy + k
            if not (0 <= x < nx and 0 <= y < ny):
                continue
            d = j * j + k * k
            if (not ring and d <= r * r) or (ring and r * r < d <= (r + 1) * (r + 1)):
                img[y, x, :3] = colour


def render_field(field, occ, source, ball):
    """saveStandAloneVisibility / saveRayCastingVisibility restated (reference solver.cpp:898-955, 960-1017): grey field
    (field row 0 is never drawn: it keeps the constructor's black, :13-33), yellow source ball with a black ring, red
    obstacles, image y = ny-1-y."""
    ny, nx = occ.shape
    img = np.zeros((ny, nx, 4), np.uint8)
    img[..., 3] = 255
    for y in range(1, ny):
        g = (255 * field[y]).astype(np.uint8)  # double -> Uint8 truncates
        img[ny - 1 - y, :, 0] = img[ny - 1 - y, :, 1] = img[ny - 1 - y, :, 2] = g
    cx, cy = source[0], ny - 1 - source[1]
    _disc(img, cx, cy, ball, YELLOW)
    _disc(img, cx, cy, ball, BLACK, ring=True)
    for y in range(1, ny):
        img[ny - 1 - y, occ[y] == 0, :3] = RED
    return img


def render_path(occ, path, ball):
    """saveImageWithPath restated (reference solver.cpp:1218-1292) over the constructor's map image (:20-33)."""
    ny, nx = occ.shape
    img = np.zeros((ny, nx, 4), np.uint8)
    img[..., 3] = 255
    for y in range(1, ny):
        img[ny - 1 - y, occ[y] >= 1, :3] = WHITE
    for (ax, ay), (bx, by) in zip(path[:-1], path[1:]):
        x0, y0, x1, y1 = int(ax), ny - 1 - int(ay), int(bx), ny - 1 - int(by)
        dx, dy = abs(x1 - x0), abs(y1 - y0)
        sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
        err = dx - dy
        while x0 != x1 or y0 != y1:
            img[y0, x0, :3] = MAGENTA
            e2 = 2 * err
            if e2 > -dy:
                err -= dy
                x0 += sx
            if e2 < dx:
                err += dx
                y0 += sy
    for x, y in path:
        _disc(img, int(x), ny - 1 - int(y), ball, CYAN)
    _disc(img, int(path[0][0]), ny - 1 - int(path[0][1]), ball, GREEN)
    _disc(img, int(path[-1][0]), ny - 1 - int(path[-1][1]), ball, RED)
    return img


def _png(tmp_path, name):
    from PIL import Image
    return np.array(Image.open(tmp_path / "output" / name).convert("RGBA"))


def _same_image(got, want, what):
    assert got.shape == want.shape, "%s: shape %r vs %r" % (what, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere((got != want).any(-1))
        y, x = bad[0]
        raise AssertionError("%s: %d pixels differ, first at (x=%d,y=%d): %r vs %r" % (what, len(bad), x, y, got[y, x], want[y, x]))


def test_mode1_files_byte_identical(tmp_path, oracle):
    nx, ny, nb, seed = 210, 160, 18, 5
    occ = oracle.generate_env(nx, ny, nb, 4, 30, 4, 30, seed)
    free = np.argwhere(occ == 1)
    (sy, sx), (ey, ex) = free[10], free[-10]
    cfg = BASE.format(mode=1, nx=nx, ny=ny, nb=nb, seed=seed, image="none", sx=sx, sy=sy, ex=ex, ey=ey, max_iter=80, thr=0.25)
    r = _run(tmp_path, cfg)
    assert r.returncode == 0, r.stderr
    want = oracle.solve(occ, (int(sx), int(sy)), (int(ex), int(ey)), 0.25, 80)
    assert want["status"] == 0
    assert _read(tmp_path, "cameFrom.txt") == host_lib.format_matrix(want["came_from"])
    assert _read(tmp_path, "VisibilityMap.txt") == host_lib.format_matrix(want["vis_global"])
    assert _read(tmp_path, "LocalVisibilityMap.txt") == host_lib.format_matrix(want["vis_local"])
    assert _read(tmp_path, "visibilityField.txt") == host_lib.format_matrix(occ.astype(np.float64))
    piv = want["pivots"][: want["n_pivots"]]
    assert _read(tmp_path, "lightSources.txt") == "".join("%d %d\n" % (x, y) for x, y in piv)
    d, _ = oracle.reconstruct_path(want["came_from"], want["pivots"], (int(ex), int(ey)))
    assert ("Path length: %g\n" % d) in r.stdout
    assert "Config file parsed successfully" in r.stdout and "Execution time in us:" in r.stdout
    assert "Density of the occupancy grid: %g%%" % ((occ == 0).sum() / occ.size * 100) in r.stdout
    # the three PNGs, pixel for pixel, against the reference's renderers restated over the ORACLE's arrays
    d, path = oracle.reconstruct_path(want["came_from"], want["pivots"], (int(ex), int(ey)))
    _same_image(_png(tmp_path, "ResultingPath.png"), render_path(occ, path, 4), "ResultingPath.png")
    sweep = oracle.sweep_full(occ, int(sx), int(sy))  # benchmark(): computeVisibility() from `start` (solver.cpp:214-224)
    _same_image(_png(tmp_path, "standAloneVisibility.png"), render_field(sweep, occ, (int(sx), int(sy)), 4), "standAloneVisibility.png")
    rays = oracle.raycast_all(occ, int(sx), int(sy))
    _same_image(_png(tmp_path, "rayCastingVisibility.png"), render_field(rays, occ, (int(sx), int(sy)), 4), "rayCastingVisibility.png")


def test_mode2_image_flip(tmp_path, oracle):
    # a PNG map: rows are written top-down (y = ny-1 .. 0), pivots get their y flipped back
    occ = maps.random_rect_map(150, 120, 14, 4, 25, 4, 25, 3)
    ny, nx = occ.shape
    host = host_lib.load()
    rgba = np.zeros((ny, nx, 4), np.uint8)
    rgba[..., 0] = np.where(occ == 1, 255, 0)
    rgba[..., 3] = 255
    png = str(tmp_path / "map.png")
    assert host.vhp_host_save_png(png.encode(), rgba.ctypes.data, nx, ny) == 0
    free = np.argwhere(occ == 1)
    (sy, sx), (ey, ex) = free[5], free[-5]
    # config coordinates are bottom-up in mode 2
    cfg = BASE.format(mode=2, nx=1, ny=1, nb=0, seed=0, image=png, sx=sx, sy=ny - 1 - sy, ex=ex, ey=ny - 1 - ey,
                      max_iter=80, thr=0.2)
    r = _run(tmp_path, cfg)
    assert r.returncode == 0, r.stderr
    want = oracle.solve(occ, (int(sx), int(sy)), (int(ex), int(ey)), 0.2, 80)
    assert want["status"] == 0
    assert _read(tmp_path, "cameFrom.txt") == host_lib.format_matrix(want["came_from"], flip=True)
    assert _read(tmp_path, "VisibilityMap.txt") == host_lib.format_matrix(want["vis_global"], flip=True)
    assert _read(tmp_path, "visibilityField.txt") == host_lib.format_matrix(occ.astype(np.float64), flip=True)
    piv = want["pivots"][: want["n_pivots"]]
    assert _read(tmp_path, "lightSources.txt") == "".join("%d %d\n" % (x, ny - 1 - y) for x, y in piv)


def _interface_m_walk(tmp_path, start_cfg, end_cfg):
    """What interface.m does with the files (reference interface.m:58-162), restated: readtable with ' ' as delimiter (a numeric
    matrix per file, one row per line), cameFrom = T + 1 (:85), pivots = T + 1 (:133), then from the end point
    pt = pivots(cameFrom(pt(2), pt(1)), :) until pt is the start point (:141-162), summing norm() of the hops.  start / end
    are the config's own numbers (+ 1): in mode 2 the files and the config share the image's y orientation."""
    def table(name):
        return np.array([[float(t) for t in line.split()] for line in (tmp_path / "output" / name).read_text().splitlines()])
    came = table("cameFrom.txt") + 1
    pivots = table("lightSources.txt") + 1
    sp, pt = np.array(start_cfg, float) + 1, np.array(end_cfg, float) + 1
    path, total = [pt.copy()], 0.0
    for _ in range(len(pivots) + 2):            # the MATLAB loop has no bound; a consistent parent table needs at most this many hops
        if pt[0] == sp[0] and pt[1] == sp[1]:
            break
        label = came[int(pt[1]) - 1, int(pt[0]) - 1]          # cameFrom(pt(2), pt(1)), 1-based
        assert 1 <= label <= len(pivots), "cameFrom + 1 = %r is not a row of pivots" % label
        pt = pivots[int(label) - 1].copy()
        total += float(np.linalg.norm(pt - path[-1]))
        path.append(pt.copy())
    assert pt[0] == sp[0] and pt[1] == sp[1], "the back-track of interface.m:141-162 does not reach the start"
    return np.array(path) - 1, total, came - 1, pivots - 1


@pytest.mark.parametrize("mode", [1, 2])
def test_files_as_interface_m_reads_them(tmp_path, oracle, mode):
    # the consumer's view of the .txt surface: interface.m's own parsing and back-track over the CLI's files terminates on the
    # start, and equals vhp_reconstruct_path over the same arrays and the "Path length" the CLI prints
    import vhp_amd
    if mode == 1:
        nx, ny, nb, seed = 210, 160, 18, 5
        occ = oracle.generate_env(nx, ny, nb, 4, 30, 4, 30, seed)
        free = np.argwhere(occ == 1)
        (sy, sx), (ey, ex) = free[10], free[-10]
        start_cfg, end_cfg = (int(sx), int(sy)), (int(ex), int(ey))
        cfg = BASE.format(mode=1, nx=nx, ny=ny, nb=nb, seed=seed, image="none", sx=sx, sy=sy, ex=ex, ey=ey, max_iter=80, thr=0.25)
    else:
        occ = maps.random_rect_map(150, 120, 14, 4, 25, 4, 25, 3)
        ny, nx = occ.shape
        host = host_lib.load()
        rgba = np.zeros((ny, nx, 4), np.uint8)
        rgba[..., 0] = np.where(occ == 1, 255, 0)
        rgba[..., 3] = 255
        png = str(tmp_path / "map.png")
        assert host.vhp_host_save_png(png.encode(), rgba.ctypes.data, nx, ny) == 0
        free = np.argwhere(occ == 1)
        (sy, sx), (ey, ex) = free[5], free[-5]
        start_cfg, end_cfg = (int(sx), int(ny - 1 - sy)), (int(ex), int(ny - 1 - ey))   # bottom-up in the config (solver.cpp:83-86)
        cfg = BASE.format(mode=2, nx=1, ny=1, nb=0, seed=0, image=png, sx=start_cfg[0], sy=start_cfg[1], ex=end_cfg[0], ey=end_cfg[1],
                          max_iter=80, thr=0.2)
    r = _run(tmp_path, cfg)
    assert r.returncode == 0, r.stderr
    path, total, came, pivots = _interface_m_walk(tmp_path, start_cfg, end_cfg)
    assert len(path) >= 2 and tuple(path[-1]) == tuple(map(float, start_cfg)) and tuple(path[0]) == tuple(map(float, end_cfg))
    # the library's own reconstruction over the arrays the files hold (the trailing `end` entry is not in lightSources.txt)
    piv_all = np.concatenate([pivots, np.array([end_cfg], float)]).astype(np.int32)
    length, lib_path = vhp_amd.reconstruct_path(came.astype(np.uint64), piv_all, end_cfg)
    assert [tuple(p) for p in lib_path.tolist()] == [tuple(int(v) for v in p) for p in path[::-1]]
    assert abs(length - total) <= 1e-9 * max(1.0, total)
    assert ("Path length: %g\n" % length) in r.stdout


def test_error_messages_and_untouched_output(tmp_path, oracle):
    occ = oracle.generate_env(100, 100, 10, 4, 30, 4, 30, 9)
    by, bx = np.argwhere(occ == 0)[0]
    fy, fx = np.argwhere(occ == 1)[0]
    r = _run(tmp_path, BASE.format(mode=1, nx=100, ny=100, nb=10, seed=9, image="none", sx=bx, sy=by, ex=fx, ey=fy,
                                   max_iter=10, thr=0.5))
    assert "Start point is not valid (occupied)" in r.stdout
    assert not (tmp_path / "output" / "cameFrom.txt").exists()  # solve() returned before saveResults()
    r = _run(tmp_path, BASE.format(mode=1, nx=100, ny=100, nb=10, seed=9, image="none", sx=fx, sy=fy, ex=100, ey=5,
                                   max_iter=10, thr=0.5))
    assert "End point is out of bounds." in r.stdout
    (tmp_path / "config" / "settings.config").write_text("ncols=abc\n")
    r = subprocess.run([host_lib.CLI], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Error parsing config file" in r.stdout


def test_benchmark_series_appends_four_columns(tmp_path, oracle):
    # benchmarkSeries() (reference solver.cpp:295-374): log-spaced empty grids 50 .. 5000, source at the centre, one line
    # "t_visibility t_raycast ratio NxN" per size APPENDED to output/benchmark_results.txt (:368-373).  Three sizes here
    # (50, 500, 5000) instead of the reference's 60; the map the solver was built on must be restored afterwards.
    occ = oracle.generate_env(120, 90, 8, 4, 30, 4, 30, 2)
    free = np.argwhere(occ == 1)
    (sy, sx), (ey, ex) = free[3], free[-3]
    cfg = BASE.format(mode=1, nx=120, ny=90, nb=8, seed=2, image="none", sx=sx, sy=sy, ex=ex, ey=ey, max_iter=80, thr=0.25)
    (tmp_path / "config").mkdir(exist_ok=True)
    (tmp_path / "config" / "settings.config").write_text(cfg)
    for run in (1, 2):
        r = subprocess.run([host_lib.CLI, "--benchmark-series", "--series-points", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        lines = (tmp_path / "output" / "benchmark_results.txt").read_text().splitlines()
        assert len(lines) == 3 * run  # appended, not overwritten
        for line, size in zip(lines[-3:], (50, 500, 5000)):
            tok = line.split()
            assert len(tok) == 4 and tok[3] == "%dx%d" % (size, size)
            t_vis, t_ray, ratio = float(tok[0]), float(tok[1]), float(tok[2])
            assert t_vis > 0 and t_ray > 0 and abs(ratio - t_ray / t_vis) <= 1e-4 * ratio + 1e-6
        for size in (50, 500, 5000):
            assert "For grid size: %dx%d" % (size, size) in r.stdout
        assert "Ratios: " in r.stdout


def test_raycast_1000_with_border_sources(oracle):
    # ray casting at BASELINE's grid size, sources on the border and in the corners (Bresenham rays that start on the
    # frame; the reference's size_t arithmetic never leaves the grid because both ends are inside)
    import torch  # noqa: F401
    import vhp_amd
    occ, _ = maps.config_c3(4)
    c = vhp_amd.Context(0)
    c.set_map(occ)
    for sx, sy in [(0, 0), (999, 999), (999, 0), (0, 500), (517, 999), (423, 311)]:
        if not occ[sy, sx]:
            continue
        got, want = c.raycast_all(sx, sy), oracle.raycast_all(occ, sx, sy)
        assert got.tobytes() == want.tobytes(), "ray casting from (%d,%d): %d cells differ" % (sx, sy, int((got != want).sum()))


def test_raycast_matches_oracle(oracle):
    import torch  # noqa: F401
    import vhp_amd
    occ = maps.random_rect_map(90, 70, 12, 3, 15, 3, 15, 4)
    c = vhp_amd.Context(0)
    c.set_map(occ)
    sx, sy = (int(v) for v in maps.free_sources(occ, 1, 2)[0])
    assert c.raycast_all(sx, sy).tobytes() == oracle.raycast_all(occ, sx, sy).tobytes()
