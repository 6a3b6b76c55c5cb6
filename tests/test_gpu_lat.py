"""The latency sweep (csrc/vhp_lat.hpp, gfx950 build) through the C ABI against the oracle, bit for bit.
Selected explicitly (vhp_set_option "kernel" = 4) on every grid it supports; the same source runs on the CPU simulator in
tests/test_lat_sim.py."""
import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def _ctx(vhp, occ):
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_option("kernel", 4)
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
                                   (640, 603), (72, 1100), (1104, 72), (1000, 1000), (1024, 700), (1016, 520), (2, 5), (10, 9), (106, 77), (130, 131), (690, 402),
                                   # odd widths (every other row starts 8 bytes off the 16-byte grid): BASELINE config 1 is 101 x 101, the
                                   # reference's benchmarkSeries sweeps round(50 * 100^(k/59)) (solver.cpp:311-324): 971, 1001 ...
                                   (1, 1), (3, 7), (9, 9), (101, 101), (105, 78), (263, 300), (689, 402), (71, 1100), (971, 971), (1001, 1001), (1051, 1000)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lat_kernel_bit_exact(vhp, oracle, nx, ny, dtype):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 4, nx + ny)
    c = _ctx(vhp, occ)
    got = c.sweep_batch(src, dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    assert c.last_sweep_kernel() == 4
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "%dx%d %s latency sweep, source (%d,%d)" % (nx, ny, dtype, sx, sy))


def test_lat_kernel_config2_and_again(vhp, oracle):
    # the C2 launch (1000 x 1000, empty, centre source) into a NaN-filled buffer, twice on one context (the scratch of the first
    # launch holds valid-looking entries of an older epoch for the second), then after a pool-sweep launch on the same scratch
    import torch
    occ = np.ones((1000, 1000), np.uint8)
    c = _ctx(vhp, occ)
    src = np.array([[500, 500]], np.int32)
    want = oracle.sweep_full(occ, 500, 500)
    d_src = torch.from_numpy(src).cuda()
    out = torch.full((1, 1000, 1000), float("nan"), dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        out.fill_(float("nan"))
        if rep == 2:
            c.set_option("kernel", 3)
            c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
            c.set_option("kernel", 4)
            out.fill_(float("nan"))
        c.sweep_batch_device(d_src.data_ptr(), 1, out.data_ptr())
        torch.cuda.synchronize()
        assert c.last_sweep_kernel() == 4
        _assert_same(out[0].cpu().numpy(), want, "C2, launch %d" % rep)


@pytest.mark.parametrize("nx,ny", [(2048, 1500), (1536, 2600), (4096, 4096), (8192, 24), (24, 8192)])
def test_lat_kernel_large_grids_in_rounds(vhp, oracle, nx, ny):
    # more strips than wavefronts: a wavefront sweeps its strips one after the other
    occ = maps.random_rect_map(nx, ny, 40, min(10, ny // 6 - 1, nx // 6 - 1) if min(nx, ny) < 64 else 10, max(nx // 6, 2), 1 if min(nx, ny) < 64 else 10, max(ny // 6, 2), nx + 3)
    src = _sources(occ, 1, ny)[:3]
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d latency sweep, source (%d,%d)" % (nx, ny, sx, sy))


def test_lat_kernel_bad_source_is_reported(vhp):
    occ = np.zeros((64, 64), np.uint8)
    c = _ctx(vhp, occ)
    with pytest.raises(Exception):
        c.sweep_batch(np.array([[5, 5], [64, 3]], np.int32))
    # ... and in a launch of more units than CUs, whose units go out in order of length (the units of a source outside the grid: last)
    src = np.array([[k % 64, (7 * k) % 64] for k in range(40)], np.int32)
    src[23] = (3, -1)
    with pytest.raises(Exception):
        c.sweep_batch(src)


def _walled(nx, ny, seed, density):
    rng = np.random.RandomState(seed)
    occ = (rng.rand(ny, nx) >= density).astype(np.uint8)
    for k in range(3):
        occ[rng.randint(0, ny), :] = 0
        occ[:, rng.randint(0, nx)] = 0
    return occ


@pytest.mark.parametrize("nx,ny,density", [(200, 163, 0.5), (328, 300, 0.2), (640, 603, 0.08), (1104, 72, 0.3), (72, 1100, 0.3), (130, 131, 0.95),
                                           (1000, 1000, 0.05), (2048, 1500, 0.02), (201, 163, 0.5), (329, 301, 0.2), (131, 130, 0.95), (1001, 999, 0.05)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lat_kernel_strips_that_die(vhp, oracle, nx, ny, density, dtype):
    # maps where the light dies early: dead strips stop sweeping and store zeros; NaN-filled output, every cell against the oracle
    import torch
    occ = _walled(nx, ny, nx + ny, density)
    src = _sources(occ, 5, nx * 3 + ny)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    tdt = torch.float64 if dtype == "f64" else torch.float32
    out = torch.full((len(src), ny, nx), float("nan"), dtype=tdt, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), len(src), out.data_ptr(), dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    torch.cuda.synchronize()
    assert c.last_sweep_kernel() == 4
    got = out.cpu().numpy()
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "%dx%d density %.2f %s, source (%d,%d)" % (nx, ny, density, dtype, sx, sy))


@pytest.mark.parametrize("nx,ny,density", [(1104, 1030, 0.0), (1104, 1030, 0.03), (2600, 1100, 0.01), (1100, 4200, 0.01), (4200, 1100, 0.0), (1031, 1101, 0.02)])
def test_lat_kernel_several_workgroups_per_unit(vhp, oracle, nx, ny, density):
    """Sides above 1024: the bands of an octant are dealt out to two, four or eight workgroups (csrc/vhp_lat.hip lat_halves; LatArgs::halves), a
    band above a band of another workgroup reads that band's line in global memory and its death from a record beside the lines.  Sources in
    the corners and in the middle (octants of one band and of sixty), walls (bands that die early, late, never), NaN-filled output, an odd
    width, both dtypes, every cell against the oracle -- and the same launch again (the records of the first one are stale tags by then)."""
    import torch
    occ = _walled(nx, ny, nx + ny, density) if density > 0 else np.ones((ny, nx), np.uint8)
    src = _sources(occ, 2, nx + ny)[:6]
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    want = [oracle.sweep_full(occ, int(sx), int(sy)) for sx, sy in src]
    for dtype in ("f64", "f32"):
        tdt = torch.float64 if dtype == "f64" else torch.float32
        for rep in range(2):
            out = torch.full((len(src), ny, nx), float("nan"), dtype=tdt, device="cuda")
            c.sweep_batch_device(d_src.data_ptr(), len(src), out.data_ptr(), dtype=vhp.F64 if dtype == "f64" else vhp.F32)
            torch.cuda.synchronize()
            assert c.last_sweep_kernel() == 4
            got = out.cpu().numpy()
            for k, (sx, sy) in enumerate(src):
                _assert_same(got[k], want[k] if dtype == "f64" else want[k].astype(np.float32), "%dx%d density %.2f %s launch %d, source (%d,%d)" % (nx, ny, density, dtype, rep, sx, sy))


@pytest.mark.parametrize("workgroups", [2, 4, 8])
def test_lat_kernel_workgroups_per_unit_option(vhp, oracle, workgroups):
    """vhp_set_option("lat_workgroups", n): the launch shape of the large grids asked for on a small one (1000 x 1000 with walls, 690 x 402),
    where most bands of a second or fourth workgroup die before, while or after their readers start; the results do not depend on it."""
    import torch
    for nx, ny, density in [(1000, 1000, 0.03), (690, 402, 0.0), (1000, 1000, 0.0)]:
        occ = _walled(nx, ny, nx + ny + workgroups, density) if density > 0 else np.ones((ny, nx), np.uint8)
        src = _sources(occ, 2, nx + ny)[:4]
        c = _ctx(vhp, occ)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_option("lat_workgroups", workgroups)
        d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
        out = torch.full((len(src), ny, nx), float("nan"), dtype=torch.float64, device="cuda")
        c.sweep_batch_device(d_src.data_ptr(), len(src), out.data_ptr())
        torch.cuda.synchronize()
        assert c.last_sweep_kernel() == 4
        got = out.cpu().numpy()
        for k, (sx, sy) in enumerate(src):
            _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d, %d workgroups per unit, source (%d,%d)" % (nx, ny, workgroups, sx, sy))


def test_lat_kernel_two_contexts_side_by_side(vhp):
    """Two contexts on two streams of one device, launches of several workgroups per unit in flight side by side (2100 x 2048: four per unit,
    2 x 8 x 4 x 8 = 512 workgroups on 256 CUs): a launch sized for the whole device gets part of it and must neither hang nor differ -- a
    unit's workgroups are next to each other in the launch and wait only for each other (tools/lat_side_by_side.py for larger grids)."""
    import torch
    nx, ny, n = 2100, 2048, 8
    occ = maps.random_rect_map(nx, ny, 40, 20, 300, 20, 300, 9)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs, outs, d_src = [], [], []
    for k in range(2):
        c = vhp.Context(0)
        c.set_stream(streams[k].cuda_stream)
        c.set_map(occ)
        ctxs.append(c)
        outs.append(torch.full((n, ny, nx), float("nan"), dtype=torch.float64, device="cuda"))
        d_src.append(torch.from_numpy(np.ascontiguousarray(maps.free_sources(occ, n, 3 + k), np.int32)).cuda())
    torch.cuda.synchronize()
    refs = []
    for k in range(2):   # each launch alone
        ctxs[k].sweep_batch_device(d_src[k].data_ptr(), n, outs[k].data_ptr())
        torch.cuda.synchronize()
        assert ctxs[k].last_sweep_kernel() == 4
        refs.append(outs[k].clone())
        outs[k].fill_(float("nan"))
    torch.cuda.synchronize()
    for it in range(5):   # ... and side by side
        for k in range(2):
            ctxs[k].sweep_batch_device(d_src[k].data_ptr(), n, outs[k].data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1])


def test_lat_kernel_maze_6_pivots(vhp, oracle):
    occ = maps.maze_6()
    res = oracle.solve(occ, (345, 391), (341, 10), 0.1, 1000)
    piv = res["pivots"][: res["n_pivots"] + 1]
    src = np.array([p for p in piv if occ[p[1], p[0]]], np.int32)[:24]
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "maze_6 pivot %d (%d,%d)" % (k, sx, sy))


def test_lat_kernel_more_units_than_cus_by_default(vhp, oracle):
    # the largest launches the library gives to the latency sweep by itself: 96 sources up to 640 cells a side = 768 workgroups on 256 CUs,
    # launched by falling length of their march (vhp_lat_order: most octants are short, and find their CUs long before the longest are through)
    occ = maps.random_rect_map(328, 300, 30, 3, 40, 3, 40, 77)
    src = maps.free_sources(occ, 96, 5)
    src[5] = (0, 0); src[17] = (327, 299); src[40] = (327, 0)
    occ[src[:, 1], src[:, 0]] = 1
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.sweep_batch(src)
    assert c.last_sweep_kernel() == 4
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "96 sources, source %d (%d,%d)" % (k, sx, sy))
    got = c.sweep_batch(np.concatenate([src, src[:1]]))   # 97: the front sweep again
    assert c.last_sweep_kernel() == 1
    # ... and 64 at 1000 x 1000 (65: the pool sweep), every field of the launch against the oracle
    occ, src = maps.config_c3(65)
    c = vhp.Context(0)
    c.set_map(occ)
    got = c.sweep_batch(src[:64])
    assert c.last_sweep_kernel() == 4
    for k, (sx, sy) in enumerate(src[:64]):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "64 sources at 1000^2, source %d (%d,%d)" % (k, sx, sy))
    c.sweep_batch(src)
    assert c.last_sweep_kernel() == 3


def test_lat_kernel_random_campaign(vhp, oracle):
    # a fixed pseudo-random campaign: sizes, obstacle kinds (rectangles / salt of three densities with walls), source counts
    rng = np.random.RandomState(20261003)
    for case in range(60):
        nx = int(rng.choice([8, 16, 40, 72, 104, 130, 200, 264, 328, 520, 690, 1000]))
        ny = int(rng.randint(1, 900))
        if rng.randint(0, 3) == 0:
            occ = maps.random_rect_map(nx, ny, int(rng.randint(1, 40)), 1, max(nx // 6, 2), 1, max(ny // 6, 2), int(rng.randint(1, 1 << 20)))
        else:
            occ = (rng.rand(ny, nx) >= [0.03, 0.15, 0.5][rng.randint(0, 3)]).astype(np.uint8)
            for k in range(rng.randint(0, 4)):
                occ[rng.randint(0, ny), :] = 0
                occ[:, rng.randint(0, nx)] = 0
        ns = int(rng.randint(1, 9))
        src = np.stack([rng.randint(0, nx, ns), rng.randint(0, ny, ns)], 1).astype(np.int32)
        occ[src[:, 1], src[:, 0]] = 1
        got = _ctx(vhp, occ).sweep_batch(src)
        for k, (sx, sy) in enumerate(src):
            _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "case %d: %dx%d, source (%d,%d)" % (case, nx, ny, sx, sy))
