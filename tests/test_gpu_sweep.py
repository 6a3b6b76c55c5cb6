"""Parity of the HIP sweep (through the C ABI) with the CPU oracle.  Bit-exact: the
kernels compute in IEEE binary64 with the reference's operation order."""
import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401  (first, so the extension shares torch's HIP runtime)
    import vhp_amd
    return vhp_amd


def _ctx(vhp, occ, **options):
    """options: launch-shape overrides through the ABI (vhp_set_option), e.g. rows_per_lane=1, strips=8, slide=1."""
    c = vhp.Context(0)
    c.set_map(occ)
    for k, v in options.items():
        c.set_option(k, v)
    return c


# the front sweep's shape for batches of 256+ sources (vhp_capi.hip pick_shape)
LINE_SHAPE = dict(rows_per_lane=1, strips=8, slide=1)


def _assert_same(got, want, what):
    if got.tobytes() != want.tobytes():
        bad = np.argwhere(got != want)
        y, x = bad[0][-2:]
        raise AssertionError("%s: %d cells differ, first at (x=%d,y=%d): got %r want %r" % (
            what, len(bad), x, y, got[tuple(bad[0])], want[tuple(bad[0])]))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_small_random_maps_bit_exact(vhp, oracle, seed):
    occ = maps.random_rect_map(101, 77, 25, 2, 20, 2, 20, seed)
    src = maps.free_sources(occ, 24, seed + 100)
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "seed %d source %d (%d,%d)" % (seed, k, sx, sy))


def test_config1_101x101(vhp, oracle):
    # BASELINE config 1: the 101x101 grid derived from MATLAB_code/rnd_1.mat (tests/golden/c1_rnd1_mask.npz),
    # single source (5,5); plus the other corner region and an LCG map of the same size
    occ = maps.c1_rnd1_mask()
    assert occ.shape == (101, 101) and occ[5, 5] == 1
    src = np.array([[5, 5], [95, 95], [4, 4], [50, 50]], np.int32)
    src = src[[bool(occ[y, x]) for x, y in src]]
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "config 1 (rnd_1 mask) source (%d,%d)" % (sx, sy))
    occ = maps.random_rect_map(101, 101, 25, 2, 20, 2, 20, 1)
    occ[5, 5] = 1
    got = _ctx(vhp, occ).sweep_batch(np.array([[5, 5]], np.int32))
    _assert_same(got[0], oracle.sweep_full(occ, 5, 5), "config 1 (LCG map)")


def test_border_blocked_and_thin(vhp, oracle):
    occ = maps.random_rect_map(40, 33, 10, 2, 8, 2, 8, 5)
    src = np.array([(0, 0), (39, 0), (0, 32), (39, 32), (0, 17), (20, 0), (39, 5), (7, 32), (1, 1)], np.int32)
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "border source (%d,%d)" % (sx, sy))
    for nx, ny in [(1, 1), (1, 9), (9, 1), (2, 2), (3, 70), (70, 3), (65, 64), (64, 65), (129, 130)]:
        occ = np.ones((ny, nx), np.uint8)
        if nx > 2 and ny > 2:
            occ[ny // 2, nx // 2] = 0
        src = np.array([(x, y) for x in {0, nx // 2, nx - 1} for y in {0, ny // 2, ny - 1}], np.int32)
        got = _ctx(vhp, occ).sweep_batch(src)
        for k, (sx, sy) in enumerate(src):
            _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d source (%d,%d)" % (nx, ny, sx, sy))


# (batches of up to 32 sources on even widths take the latency sweep by default: kernel = 1 keeps the front sweep's default shapes covered)
@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("side", [200, 300, 640])
def test_multi_strip_sizes(vhp, oracle, side, kernel):
    # sides that need 1, several and many wavefront strips per octant
    occ = maps.random_rect_map(side, side - 37, 30, 3, side // 8, 3, side // 8, side)
    src = maps.free_sources(occ, 6, side)
    got = _ctx(vhp, occ, kernel=kernel).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "side %d source %d" % (side, k))


@pytest.mark.parametrize("nx,ny", [(101, 77), (300, 263), (1000, 1000), (1500, 1100)])
def test_slid_column_grid_bit_exact(vhp, oracle, nx, ny):
    # large batches slide the y-major column grid onto 128-byte lines (strip 0 then owns columns that do
    # not exist); force that path for a small batch and compare cell by cell
    occ = maps.random_rect_map(nx, ny, 30, 3, max(nx // 8, 4), 3, max(ny // 8, 4), nx + ny)
    src = np.concatenate([maps.free_sources(occ, 5, nx), np.array([(0, 0), (nx - 1, ny - 1), (nx - 1, 0), (1, ny - 2)], np.int32)])
    occ[src[:, 1], src[:, 0]] = 1
    got = _ctx(vhp, occ, slide=1).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d slid grid, source (%d,%d)" % (nx, ny, sx, sy))


@pytest.mark.parametrize("nx,ny", [(1000, 1000), (1024, 700), (640, 603), (1016, 520), (4096, 300), (300, 2048)])
def test_line_mode_shape_bit_exact(vhp, oracle, nx, ny):
    # batches of 256+ sources on sides above 256 run in the one-row-per-lane shape, whose x-major strips hold rows
    # back and flush whole 128-byte lines (rows alternate between the two line phases when nx/8 is odd); force
    # that shape, and the slid column grid that goes with large batches, for a small batch
    occ = maps.random_rect_map(nx, ny, 40, 5, nx // 8, 5, ny // 8, nx * 3 + ny)
    src = np.concatenate([maps.free_sources(occ, 4, ny), np.array([(0, 0), (nx - 1, ny - 1), (nx - 2, 1), (7, ny - 8)], np.int32)])
    occ[src[:, 1], src[:, 0]] = 1
    got = _ctx(vhp, occ, **LINE_SHAPE).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d line mode, source (%d,%d)" % (nx, ny, sx, sy))


@pytest.mark.parametrize("kernel", [0, 1])
def test_config3_1000x1000_subset(vhp, oracle, kernel):
    occ, src = maps.config_c3(256)
    pick = src[[0, 17, 101, 255]]
    extra = np.array([[0, 0], [999, 999], [999, 0], [500, 500]], np.int32)
    extra = extra[[bool(occ[y, x]) for x, y in extra]]
    pick = np.concatenate([pick, extra])
    got = _ctx(vhp, occ, kernel=kernel).sweep_batch(pick)
    for k, (sx, sy) in enumerate(pick):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "C3 source (%d,%d)" % (sx, sy))


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("nx,ny", [(1500, 1100), (2500, 2300), (2049, 700)])
def test_large_grids_multi_round(vhp, oracle, nx, ny, kernel):
    # sides above 1024 use 4 rows per lane; above 2048 the strips of an octant are swept in rounds
    occ = maps.random_rect_map(nx, ny, 40, 10, nx // 6, 10, ny // 6, nx)
    src = maps.free_sources(occ, 3, ny)
    corner = np.array([[0, 0], [nx - 1, ny - 1], [nx - 1, 0]], np.int32)
    corner = corner[[bool(occ[y, x]) for x, y in corner]]
    src = np.concatenate([src, corner])
    got = _ctx(vhp, occ, kernel=kernel).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d source (%d,%d)" % (nx, ny, sx, sy))


def test_config5_4096_subset(vhp, oracle):
    # BASELINE config 5 map (4096x4096, 50 obstacles scaled x4), a few of its sources through the host-buffer entry
    # point (a batch this small takes the front sweep; the 128-source launch is test_config5_the_launch_that_ships)
    occ, src = maps.config_c5(128)
    pick = np.concatenate([src[[0, 77]], np.array([[0, 0]], np.int32) if occ[0, 0] else src[[5]]])
    got = _ctx(vhp, occ).sweep_batch(pick)
    for k, (sx, sy) in enumerate(pick):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "C5 source (%d,%d)" % (sx, sy))


@pytest.mark.parametrize("kernel", [0, 1])
def test_config2_empty_1000(vhp, kernel):
    # README benchmark case: empty grid, centre source -> everything visible, except the
    # never-swept row 0 / column 0
    occ = np.ones((1000, 1000), np.uint8)
    got = _ctx(vhp, occ, kernel=kernel).sweep_batch(np.array([[500, 500]], np.int32))[0]
    assert (got[1:, 1:] == 1.0).all() and not got[0].any() and not got[:, 0].any()


@pytest.mark.parametrize("kernel", [0, 1])
def test_fp32_storage_is_rounded_fp64(vhp, oracle, kernel):
    occ = maps.random_rect_map(300, 257, 30, 3, 40, 3, 40, 9)
    src = maps.free_sources(occ, 5, 9)
    got = _ctx(vhp, occ, kernel=kernel).sweep_batch(src, dtype=vhp.F32)
    assert got.dtype == np.float32
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)).astype(np.float32), "fp32 source %d" % k)


@pytest.mark.parametrize("nx,ny", [(640, 603), (1000, 520), (328, 300)])
def test_fp32_large_batch_shape(vhp, oracle, nx, ny):
    # fp32 fields in the large-batch shape: the x-major strips hold rows back and flush whole 64-byte sectors
    occ = maps.random_rect_map(nx, ny, 30, 4, nx // 8, 4, ny // 8, nx + 7 * ny)
    src = np.concatenate([maps.free_sources(occ, 4, nx), np.array([(0, 0), (nx - 1, ny - 1)], np.int32)])
    occ[src[:, 1], src[:, 0]] = 1
    got = _ctx(vhp, occ, **LINE_SHAPE).sweep_batch(src, dtype=vhp.F32)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)).astype(np.float32), "%dx%d fp32 source (%d,%d)" % (nx, ny, sx, sy))


def test_device_resident_batch_and_full_size_properties(vhp):
    import torch
    occ, src = maps.config_c3(256)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(src).cuda()
    d_out = torch.full((len(src), 1000, 1000), -1.0, dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    assert c.last_elapsed_ms() > 0
    # size-independent properties over the whole batch
    assert float(d_out.min()) == 0.0 and float(d_out.max()) == 1.0      # every cell written, in [0,1]
    blocked = torch.from_numpy(occ == 0).cuda()
    assert float(d_out[:, blocked].abs().max()) == 0.0                  # blocked cells are dark
    sx, sy = d_src[:, 0].long(), d_src[:, 1].long()
    idx = torch.arange(len(src), device="cuda")
    assert bool((d_out[idx, sy, sx] == 1.0).all())                      # the source sees itself
    assert float(d_out[:, 0, :].abs().max()) == 0.0 and float(d_out[:, :, 0].abs().max()) == 0.0  # Q2
    # the batch ran in the large-batch launch shape (one row per lane, whole-line flushes, slid column grid);
    # the same sources as a small batch run in the small-batch shape: the fields must be the same bytes
    small = c.sweep_batch(src[:24])
    assert np.array_equal(d_out[:24].cpu().numpy(), small)
    # idempotence: a second launch into the same buffer gives the same bytes
    first = d_out.clone()
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    assert torch.equal(first, d_out)


def _device_launch(vhp, c, src, shape, dtype_t, dtype_v):
    import torch
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    d_out = torch.full((len(src),) + shape, -1.0, dtype=dtype_t, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr(), dtype=dtype_v)
    c.sync()
    return d_out


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("kernel", [0, 1, 3])
def test_config3_all_256_fields_of_the_bench_launch(vhp, oracle, dtype, kernel):
    # the launch bench.py times (256 C3 sources, device-resident, default shape selection): EVERY field against
    # the oracle, cell by cell.  kernel 0 = what the library picks by itself (the pool sweep for a batch this size),
    # 1 = the front sweep in its large-batch shape, 3 = pool sweep
    import torch
    occ, src = maps.config_c3(256)
    c = _ctx(vhp, occ, kernel=kernel)
    d_out = _device_launch(vhp, c, src, occ.shape, torch.float64 if dtype == "f64" else torch.float32,
                           vhp.F64 if dtype == "f64" else vhp.F32)
    for lo in range(0, 256, 32):
        got = d_out[lo: lo + 32].cpu().numpy()
        for k in range(32):
            sx, sy = int(src[lo + k][0]), int(src[lo + k][1])
            want = oracle.sweep_full(occ, sx, sy)
            _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "C3 %s bench launch, source %d (%d,%d)" % (dtype, lo + k, sx, sy))


@pytest.mark.parametrize("shape", [dict(), dict(pack=1), LINE_SHAPE, dict(rows_per_lane=2, strips=4, multi_round=1)])
def test_front_sweep_every_shape_with_border_sources(vhp, oracle, shape):
    # every compiled shape of the front sweep on one batch: 40 sources on a 1000x1000 map, incl. corner and border
    # sources (empty quadrants, packed short ones)
    occ, src = maps.config_c3(34)
    src = np.concatenate([src, np.array([[0, 0], [999, 999], [999, 0], [0, 999], [500, 0], [0, 500]], np.int32)])
    occ = occ.copy()
    occ[src[:, 1], src[:, 0]] = 1
    got = _ctx(vhp, occ, kernel=1, **shape).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%r source (%d,%d)" % (shape, sx, sy))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_4096_large_batch_shape_eight_rounds(vhp, oracle, dtype):
    # what 256+ sources run in at 4096^2: one row per lane, 8 strips per round => 8 rounds per octant, whole-line
    # flushes, slid column grid.  Forced through the ABI options for a handful of sources incl. the corners.
    occ, src = maps.config_c5(128)
    pick = np.concatenate([src[[3, 64]], np.array([[0, 0], [4095, 4095], [4095, 0], [1, 4094]], np.int32)])
    pick = pick[[bool(occ[y, x]) for x, y in pick]]
    got = _ctx(vhp, occ, kernel=1, **LINE_SHAPE).sweep_batch(pick, dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    for k, (sx, sy) in enumerate(pick):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "4096^2 %s 8-round shape, source (%d,%d)" % (dtype, sx, sy))


def _shipping_launch(vhp, oracle, occ, src, what, every):
    """The launch bench.py times for this workload -- all sources of rank 0, device-resident, the library's own choice of
    kernel and shape -- on a NaN-filled output: every cell of every field must be written, a second launch must leave
    the same bytes, and every `every`-th field must equal the oracle's cell for cell.  Returns the kernel that ran."""
    import torch
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    d_out = torch.full((len(src),) + occ.shape, float("nan"), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    kernel = c.last_sweep_kernel()
    for lo in range(0, len(src), 16):                              # in slices: isnan() of 17 GB at once needs another 2 GB
        assert not bool(torch.isnan(d_out[lo: lo + 16]).any()), "%s: cells left unwritten in fields %d.." % (what, lo)
    picks = list(range(0, len(src), every))
    first = d_out[picks].clone()
    for n, k in enumerate(picks):
        sx, sy = int(src[k][0]), int(src[k][1])
        _assert_same(first[n].cpu().numpy(), oracle.sweep_full(occ, sx, sy), "%s (kernel %d), source %d (%d,%d)" % (what, kernel, k, sx, sy))
    checks = [d_out[k].sum().item() for k in range(len(src))]
    c.sweep_batch_device(d_src.data_ptr(), len(src), d_out.data_ptr())
    c.sync()
    assert c.last_sweep_kernel() == kernel
    assert torch.equal(first, d_out[picks])
    assert checks == [d_out[k].sum().item() for k in range(len(src))]
    return kernel


def test_config5_the_launch_that_ships(vhp, oracle):
    # bench.py --workload c5: 128 rank-0 sources at 4096^2 (17 GB of fields), default kernel choice -- the loaded launch,
    # with its queue contended and every kind of unit side by side; 16 of its fields (every 8th) against the oracle
    occ, src = maps.config_c5(128)
    kernel = _shipping_launch(vhp, oracle, occ, src, "C5 bench launch", 8)
    assert kernel == 3  # the pool sweep (vhp_capi.hip use_pool_kernel): a threshold that moves must not silently change what this covers


def test_3072_sixtyfour_sources_the_launch_that_ships(vhp, oracle):
    # the other threshold of the kernel choice (vhp_capi.hip use_pool_kernel): 64 sources from 3072 up
    occ = maps.random_rect_map(3072, 3072, 50, 60, 300, 60, 300, seed=2)
    src = maps.free_sources(occ, 64, seed=13)
    assert _shipping_launch(vhp, oracle, occ, src, "3072^2 x 64", 8) == 3


def test_config5_fronts_sixteen_sources(vhp, oracle):
    # the front sweep's own 4096^2 shape (four rows per lane, slid grid) on 16 of the C5 sources: what a batch below the
    # pool sweep's threshold runs in
    import torch
    occ, src = maps.config_c5(128)
    pick = src[::8]
    c = _ctx(vhp, occ, kernel=1, slide=1)
    d_out = _device_launch(vhp, c, pick, occ.shape, torch.float64, vhp.F64)
    assert c.last_sweep_kernel() == 1
    for k, (sx, sy) in enumerate(pick):
        _assert_same(d_out[k].cpu().numpy(), oracle.sweep_full(occ, int(sx), int(sy)), "C5 front sweep, source %d (%d,%d)" % (8 * k, sx, sy))


@pytest.mark.parametrize("nx,ny", [(1500, 1100), (2500, 2300)])
@pytest.mark.parametrize("shape", [dict(rows_per_lane=2, strips=4, multi_round=1), dict(rows_per_lane=4, strips=8),
                                   dict(rows_per_lane=4, strips=4, multi_round=1), dict(rows_per_lane=2, strips=8, multi_round=1, slide=1)])
def test_fp32_forced_shapes_above_1024(vhp, oracle, nx, ny, shape):
    # fp32 fields in the two- and four-rows-per-lane instantiations, single- and multi-round, on sides above 1024
    occ = maps.random_rect_map(nx, ny, 40, 10, nx // 6, 10, ny // 6, nx + 1)
    src = np.concatenate([maps.free_sources(occ, 3, ny + 1), np.array([(0, 0), (nx - 1, ny - 1)], np.int32)])
    occ[src[:, 1], src[:, 0]] = 1
    got = _ctx(vhp, occ, **shape).sweep_batch(src, dtype=vhp.F32)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)).astype(np.float32), "%dx%d fp32 %r source (%d,%d)" % (nx, ny, shape, sx, sy))


def test_two_contexts_and_caller_device_untouched(vhp, oracle):
    # the dynamic-LDS limit is raised per context (per device), and no entry point changes the caller's device
    import torch
    before = torch.cuda.current_device()
    occ = maps.random_rect_map(300, 263, 20, 3, 40, 3, 40, 4)
    src = maps.free_sources(occ, 3, 4)
    a, b = _ctx(vhp, occ), _ctx(vhp, occ)
    ga, gb = a.sweep_batch(src), b.sweep_batch(src)
    assert torch.cuda.current_device() == before
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(ga[k], want, "context a")
        _assert_same(gb[k], want, "context b")


def test_errors(vhp):
    c = vhp.Context(0)
    with pytest.raises(vhp.VhpError) as e:
        c.sweep_batch(np.array([[0, 0]], np.int32))
    assert e.value.code == vhp.VHP_ERR_NO_MAP
    c.set_map(np.ones((8, 8), np.uint8))
    with pytest.raises(vhp.VhpError) as e:
        c.sweep_batch(np.array([[8, 0]], np.int32))
    assert e.value.code == vhp.VHP_ERR_SOURCE_OOB
    assert c.sweep_batch(np.zeros((0, 2), np.int32)).shape == (0, 8, 8)
    with pytest.raises(vhp.VhpError) as e:
        c.set_option("no_such_option", 1)
    assert e.value.code == vhp.VHP_ERR_ARG
    with pytest.raises(vhp.VhpError):
        c.set_option("rows_per_lane", 3)
