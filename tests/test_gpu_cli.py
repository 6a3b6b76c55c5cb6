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
    for png in ("ResultingPath.png", "standAloneVisibility.png", "rayCastingVisibility.png"):
        assert (tmp_path / "output" / png).stat().st_size > 100


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


def test_raycast_matches_oracle(oracle):
    import torch  # noqa: F401
    import vhp_amd
    occ = maps.random_rect_map(90, 70, 12, 3, 15, 3, 15, 4)
    c = vhp_amd.Context(0)
    c.set_map(occ)
    sx, sy = (int(v) for v in maps.free_sources(occ, 1, 2)[0])
    assert c.raycast_all(sx, sy).tobytes() == oracle.raycast_all(occ, sx, sy).tobytes()
