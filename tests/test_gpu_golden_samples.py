"""HIP path against golden outputs the REFERENCE ITSELF ships: Samples/SFMLrayCastingVisibility.png and
Samples/SFMLstandAloneVisibility.png (README.md:27-33; committed as data in tests/golden/samples_1000.npz by
tests/golden/make_fixtures.py).  The CPU side of the same pin is tests/test_oracle_kat.py::test_reference_sample_*.

The published sweep image was rendered by a build whose `offset` local (src/visibilityBasedSolver.cpp:573) was 1.0; HEAD
has 0.0.  So: ray casting is compared outright (HIP kernel + the CLI's renderer == the reference's PNG on every one of
its 10^6 pixels); the sweep image is reproduced pixel for pixel by vhp_sweep_batch_offset(offset = 1), and the tuned
HEAD-semantics kernels (offset = 0) are compared with the oracle, bit for bit, on the same decoded map.
"""
import os
import subprocess

import numpy as np
import pytest

import host_lib
import maps
from test_gpu_cli import BASE, _png, _same_image, render_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return maps.samples_1000()


@pytest.fixture(scope="module")
def ctx(gold):
    import torch  # noqa: F401
    import vhp_amd
    c = vhp_amd.Context(0)
    c.set_map(gold["occ"])
    return c


def _rgba(rgb):
    out = np.empty(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = rgb
    out[..., 3] = 255
    return out


def test_hip_raycasting_equals_reference_png(ctx, gold, oracle):
    rays = ctx.raycast_all(*gold["source"])
    m = gold["comparable"]
    assert int(((255 * rays).astype(np.uint8)[m] != gold["ray_grey"][m]).sum()) == 0
    assert rays.tobytes() == oracle.raycast_all(gold["occ"], *gold["source"]).tobytes()
    # the whole image, ball, ring, obstacles and the undrawn row included, through the reference's renderer restated
    _same_image(render_field(rays, gold["occ"], gold["source"], gold["ball_radius"]), _rgba(gold["ray_rgb"]), "rayCastingVisibility (python renderer)")


def test_hip_offset1_sweep_equals_reference_png(ctx, gold, oracle):
    v1 = ctx.sweep_batch_offset([gold["source"]], 1.0)[0]
    _same_image(render_field(v1, gold["occ"], gold["source"], gold["ball_radius"]), _rgba(gold["sweep_rgb"]), "standAloneVisibility at offset = 1")
    assert v1.tobytes() == oracle.sweep_full_offset(gold["occ"], *gold["source"], 1.0).tobytes()


def test_hip_head_sweep_on_reference_map(ctx, gold, oracle):
    # HEAD semantics (offset = 0): the tuned kernels == the oracle == the offset kernel at 0, and NOT the published image
    want = oracle.sweep_full(gold["occ"], *gold["source"])
    for kernel in (0, 1, 3):
        ctx.set_option("kernel", kernel)
        got = ctx.sweep_batch([gold["source"]])[0]
        assert got.tobytes() == want.tobytes(), "kernel option %d" % kernel
    ctx.set_option("kernel", 0)
    assert ctx.sweep_batch_offset([gold["source"]], 0.0)[0].tobytes() == want.tobytes()
    m = gold["comparable"]
    assert int(((255 * want).astype(np.uint8)[m] != gold["sweep_grey"][m]).sum()) > 100000
    # a batch from other sources of the same map through both kernels
    src = maps.free_sources(gold["occ"], 6, seed=5)
    src[0] = (1, 998)
    if not gold["occ"][998, 1]:
        src[0] = gold["source"]
    for kernel in (1, 3):
        ctx.set_option("kernel", kernel)
        got = ctx.sweep_batch(src)
        for k, (sx, sy) in enumerate(src):
            assert got[k].tobytes() == oracle.sweep_full(gold["occ"], int(sx), int(sy)).tobytes(), (kernel, k)
    ctx.set_option("kernel", 0)


def test_cli_renders_the_reference_raycasting_png(tmp_path, gold):
    """End to end: map PNG in (mode 2), `vhp` (solve + benchmark()), output/rayCastingVisibility.png == the reference's own
    Samples/SFMLrayCastingVisibility.png on all 1000 x 1000 pixels.  benchmark() takes `start` unflipped (:196) while
    solve() flips y in mode 2 (:83-86); both (500, 500) and (500, 499) are free.  In mode 2 field(x, y) = pixel(x, y), so
    the map image holds field row y in image row y."""
    occ = gold["occ"]
    ny, nx = occ.shape
    host = host_lib.load()
    rgba = np.zeros((ny, nx, 4), np.uint8)
    rgba[..., 0] = np.where(occ == 1, 255, 0)
    rgba[..., 3] = 255
    png = str(tmp_path / "map.png")
    assert host.vhp_host_save_png(png.encode(), rgba.ctypes.data, nx, ny) == 0
    sx, sy = gold["source"]
    assert occ[ny - 1 - sy, sx] and occ[sy, sx]
    ey, ex = [(y, x) for y, x in np.argwhere(occ == 1)[::9973] if occ[ny - 1 - y, x]][3]
    cfg = BASE.format(mode=2, nx=1, ny=1, nb=0, seed=0, image=png, sx=sx, sy=sy, ex=ex, ey=ey, max_iter=2, thr=0.25)
    cfg = cfg.replace("ballRadius=4", "ballRadius=%d" % gold["ball_radius"])
    if not os.path.exists(host_lib.CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(host_lib.PKG, "host")])
    (tmp_path / "config").mkdir(exist_ok=True)
    (tmp_path / "config" / "settings.config").write_text(cfg)
    r = subprocess.run([host_lib.CLI], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same_image(_png(tmp_path, "rayCastingVisibility.png"), _rgba(gold["ray_rgb"]), "output/rayCastingVisibility.png vs the reference's sample")
    # the sweep image of the HEAD semantics shares everything but the grey levels with the published one
    got, want = _png(tmp_path, "standAloneVisibility.png"), _rgba(gold["sweep_rgb"])
    not_grey = (want[..., 0] != want[..., 1]) | (want[..., 1] != want[..., 2])
    assert np.array_equal(got[not_grey], want[not_grey])
    assert np.array_equal(got[ny - 1], want[ny - 1]) and np.array_equal(got[:, 0], want[:, 0])  # undrawn row, unswept column
