"""The streaming sweep kernel (csrc/vhp_stream.hpp, gfx950 build) through the C ABI against the oracle, bit for bit.
Since round 3 no batch takes this kernel by default (the pool sweep, tests/test_gpu_pool.py, took its place:
vhp_capi.hip use_pool_kernel); here it is selected explicitly (vhp_set_option "kernel" = 2) so that small batches exercise it too.  The same source runs on the CPU simulator in tests/test_stream_sim.py."""
import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def _ctx(vhp, occ, tile_slots=0):
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_option("kernel", 2)
    if tile_slots:
        c.set_option("stream_tile_slots", tile_slots)
    return c


def _assert_same(got, want, what):
    if got.tobytes() != want.tobytes():
        bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
        y, x = bad[0][-2:]
        raise AssertionError("%s: %d cells differ, first at (x=%d,y=%d): got %r want %r" % (what, len(bad), x, y, got[tuple(bad[0])], want[tuple(bad[0])]))


def _sources(occ, n, seed):
    ny, nx = occ.shape
    src = list(map(tuple, maps.free_sources(occ, n, seed)))
    src += [(0, 0), (nx - 1, ny - 1), (nx - 1, 0), (0, ny - 1), (min(1, nx - 1), max(ny - 2, 0)), (nx // 2, 0), (0, ny // 2)]
    src = np.array(sorted(set(src)), np.int32)
    occ[src[:, 1], src[:, 0]] = 1
    return src


@pytest.mark.parametrize("nx,ny", [(8, 8), (8, 1), (16, 3), (8, 200), (264, 9), (104, 77), (96, 96), (200, 163), (328, 300),
                                   (640, 603), (72, 1100), (1104, 72), (1000, 1000), (1024, 700), (1016, 520)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_stream_kernel_bit_exact(vhp, oracle, nx, ny, dtype):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 4, nx + ny)
    got = _ctx(vhp, occ).sweep_batch(src, dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "%dx%d %s stream, source (%d,%d)" % (nx, ny, dtype, sx, sy))


@pytest.mark.parametrize("nx,ny", [(264, 9), (200, 163), (640, 603), (1000, 1000), (1024, 700)])
@pytest.mark.parametrize("slots", [2, 4, 6, 8])
def test_stream_kernel_tile_slots(vhp, oracle, nx, ny, slots):
    # the staging tiles of x-major strips with two windows (a plain hand-off: the sweeping wavefront hands every window
    # to its flusher before it starts the next) and with four, six, eight (what grids above ~2000 cells a side get,
    # with one workgroup per CU) instead of three
    occ = maps.random_rect_map(nx, ny, 30, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 5 + ny)
    src = _sources(occ, 4, nx + 2 * ny)
    got = _ctx(vhp, occ, tile_slots=slots).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d %d-slot stream, source (%d,%d)" % (nx, ny, slots, sx, sy))


@pytest.mark.parametrize("nx,ny", [(2048, 1500), (1536, 2600), (4096, 4096), (3000, 2504), (8192, 24), (24, 8192)])
def test_stream_kernel_large_grids(vhp, oracle, nx, ny):
    # sides above 1024: more rounds per octant, larger round-to-round rows, y-major units with the whole workgroup, two tile
    # slots and (3000 x 2504) a diagonal array that wraps; the thin ones reach the largest side the library takes
    occ = maps.random_rect_map(nx, ny, 40, min(10, ny // 6 - 1, nx // 6 - 1) if min(nx, ny) < 64 else 10, max(nx // 6, 2), 1 if min(nx, ny) < 64 else 10, max(ny // 6, 2), nx + 3)
    src = _sources(occ, 2, ny)[:6]
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d stream, source (%d,%d)" % (nx, ny, sx, sy))


def test_stream_unsupported_pitch_falls_back_to_fronts(vhp, oracle):
    # a row pitch that is not a multiple of 8 cells cannot be emitted in whole sectors: the front sweep takes it
    occ = maps.random_rect_map(101, 77, 25, 2, 20, 2, 20, 1)
    src = maps.free_sources(occ, 5, 3)
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "101x77 source %d" % k)


def test_stream_rejects_bad_source_and_is_idempotent(vhp):
    import torch
    occ, src = maps.config_c3(256)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(src).cuda()
    d_out = torch.full((len(src), 1000, 1000), float("nan"), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    assert not bool(torch.isnan(d_out).any())                      # every cell of every field is written
    first = d_out.clone()
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    assert torch.equal(first, d_out)
    bad = d_src.clone()
    bad[3, 0] = 1000
    c.sweep_batch_device(bad.data_ptr(), len(src), d_out.data_ptr())
    with pytest.raises(vhp.VhpError) as e:
        c.sync()
    assert e.value.code == vhp.VHP_ERR_SOURCE_OOB


def test_kernel_choice_is_reported(vhp):
    # which kernel a batch takes is the library's decision (vhp_capi.hip use_lat_kernel / use_pool_kernel / use_stream_kernel); it
    # tells through the ABI.  1 = front sweep, 2 = streaming sweep, 3 = pool sweep, 4 = latency sweep (up to 32 sources: an octant
    # per CU)
    occ = np.ones((8, 1104), np.uint8)   # a side above 1024
    src = np.array([[k, 3] for k in range(96)], np.int32)
    c = vhp.Context(0)
    c.set_map(occ)
    assert c.last_sweep_kernel() == 0
    c.sweep_batch(src)
    assert c.last_sweep_kernel() == 3
    c.sweep_batch(src[:95])
    assert c.last_sweep_kernel() == 1
    c.sweep_batch(src[:32])
    assert c.last_sweep_kernel() == 4
    c.sweep_batch(src[:33])
    assert c.last_sweep_kernel() == 1
    c.set_option("kernel", 2)
    c.sweep_batch(src[:3])
    assert c.last_sweep_kernel() == 2
    c.set_option("kernel", 3)
    c.sweep_batch(src[:3])
    assert c.last_sweep_kernel() == 3
    occ = np.ones((40, 1000), np.uint8)  # up to 1024: the front sweep below 192 sources, the pool sweep from there
    c = vhp.Context(0)
    c.set_map(occ)
    c.sweep_batch(np.array([[k, 3] for k in range(191)], np.int32))
    assert c.last_sweep_kernel() == 1
    c.sweep_batch(np.array([[k, 3] for k in range(300)], np.int32))
    assert c.last_sweep_kernel() == 3
    occ = np.ones((40, 1001), np.uint8)  # a pitch that is not a multiple of 8 cells: always the front sweep
    c = vhp.Context(0)
    c.set_map(occ)
    c.sweep_batch(np.array([[k, 3] for k in range(300)], np.int32))
    assert c.last_sweep_kernel() == 1
    c.sweep_batch(np.array([[5, 3]], np.int32))  # (an odd width: no latency sweep either)
    assert c.last_sweep_kernel() == 1
