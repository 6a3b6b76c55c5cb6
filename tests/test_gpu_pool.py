"""The pool sweep kernel (csrc/vhp_pool.hpp, gfx950 build) through the C ABI against the oracle, bit for bit.
Selected explicitly (vhp_set_option "kernel" = 3) so that small batches exercise it; the same source runs on the CPU
simulator in tests/test_pool_sim.py."""
import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def _ctx(vhp, occ, contexts=0):
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_option("kernel", 3)
    if contexts:
        c.set_option("pool_contexts", contexts)
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
                                   (640, 603), (72, 1100), (1104, 72), (1000, 1000), (1024, 700), (1016, 520),
                                   # widths that are not a multiple of 8 (the ANYW build), even and odd
                                   (1, 1), (1, 70), (3, 2), (7, 9), (12, 9), (10, 140), (101, 101), (202, 163), (333, 77), (37, 300),
                                   (690, 402), (1101, 70), (1002, 700), (1001, 971)])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_pool_kernel_bit_exact(vhp, oracle, nx, ny, dtype):
    nb = max(3, min(40, nx * ny // 400))
    occ = maps.random_rect_map(nx, ny, nb, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx * 7 + ny)
    src = _sources(occ, 4, nx + ny)
    c = _ctx(vhp, occ)
    got = c.sweep_batch(src, dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    assert c.last_sweep_kernel() == 3
    for k, (sx, sy) in enumerate(src):
        want = oracle.sweep_full(occ, int(sx), int(sy))
        _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "%dx%d %s pool, source (%d,%d)" % (nx, ny, dtype, sx, sy))


@pytest.mark.parametrize("contexts", [1, 2, 8, 11])
def test_pool_kernel_context_counts(vhp, oracle, contexts):
    occ = maps.random_rect_map(640, 603, 30, 1, 80, 1, 75, 5 * 640 + 603)
    src = _sources(occ, 9, 640 + 2 * 603)
    got = _ctx(vhp, occ, contexts).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%d contexts, source (%d,%d)" % (contexts, sx, sy))


@pytest.mark.parametrize("nx,ny", [(2048, 1500), (1536, 2600), (4096, 4096), (3000, 2504), (8192, 24), (24, 8192), (2049, 1500), (3002, 2501), (8191, 24)])
def test_pool_kernel_large_grids(vhp, oracle, nx, ny):
    occ = maps.random_rect_map(nx, ny, 40, min(10, ny // 6 - 1, nx // 6 - 1) if min(nx, ny) < 64 else 10, max(nx // 6, 2), 1 if min(nx, ny) < 64 else 10, max(ny // 6, 2), nx + 3)
    src = _sources(occ, 2, ny)[:6]
    got = _ctx(vhp, occ).sweep_batch(src)
    for k, (sx, sy) in enumerate(src):
        _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)), "%dx%d pool, source (%d,%d)" % (nx, ny, sx, sy))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_pool_kernel_all_256_fields_of_the_bench_launch(vhp, oracle, dtype):
    # the C3 launch bench.py times, through the pool sweep: EVERY field against the oracle, cell by cell; NaN-filled
    # output (every cell written), a second launch leaves the same bytes, a bad source is reported
    import torch
    occ, src = maps.config_c3(256)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(src).cuda()
    tdt = torch.float64 if dtype == "f64" else torch.float32
    d_out = torch.full((256, 1000, 1000), float("nan"), dtype=tdt, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), 256, d_out.data_ptr(), dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    c.sync()
    assert c.last_sweep_kernel() == 3
    assert not bool(torch.isnan(d_out).any())
    first = d_out.clone()
    c.sweep_batch_device(d_src.data_ptr(), 256, d_out.data_ptr(), dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    c.sync()
    assert torch.equal(first, d_out)
    for lo in range(0, 256, 32):
        got = first[lo: lo + 32].cpu().numpy()
        for k in range(32):
            sx, sy = int(src[lo + k][0]), int(src[lo + k][1])
            want = oracle.sweep_full(occ, sx, sy)
            _assert_same(got[k], want if dtype == "f64" else want.astype(np.float32), "C3 %s pool launch, source %d (%d,%d)" % (dtype, lo + k, sx, sy))
    bad = d_src.clone()
    bad[3, 0] = 1000
    c.sweep_batch_device(bad.data_ptr(), 256, d_out.data_ptr(), dtype=vhp.F64 if dtype == "f64" else vhp.F32)
    with pytest.raises(vhp.VhpError) as e:
        c.sync()
    assert e.value.code == vhp.VHP_ERR_SOURCE_OOB


def test_probe_stores_measures_and_validates(vhp):
    # vhp_probe_stores (the measurement aid bench.py reports the state of its timed buffer with): plausible rates, the whole-line
    # pattern not slower than the split-line one, the buffer overwritten with zeros where it was used, bad arguments refused
    import torch
    occ, _ = maps.config_c3(1)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    buf = torch.full((32, 1000, 1000), 7.0, dtype=torch.float64, device="cuda")
    whole, split = c.probe_stores(buf.data_ptr(), buf.numel() * 8)
    torch.cuda.synchronize()
    assert 0.3 < split < 8.0 and 0.3 < whole < 8.0, (whole, split)   # (256 MB keep a fraction of the chip busy: plausibility only)
    assert float(buf[0, 0, 0]) == 0.0 and float(buf[31, 999, 500]) == 0.0
    with pytest.raises(vhp.VhpError):
        c.probe_stores(buf.data_ptr(), 1 << 20)          # too small to say anything
    with pytest.raises(vhp.VhpError):
        c.probe_stores(buf.data_ptr() + 8, buf.numel() * 8 - 8)   # not on a 128-byte line


@pytest.mark.parametrize("kernel", [1, 3, 4])
@pytest.mark.parametrize("nx,ny,pad", [(200, 163, 3), (200, 163, 8), (101, 77, 1), (1002, 300, 5), (208, 90, 1), (1008, 120, 6)])
def test_field_stride_pads_between_fields(vhp, oracle, kernel, nx, ny, pad):
    # "field_stride": the fields of a device batch `pad` elements apart (an odd pad puts every other field off the 16-byte grid: the
    # kernels' builds for unaligned pairs); every field bit-exact, the padding untouched
    import torch
    occ = maps.random_rect_map(nx, ny, 20, 1, max(nx // 8, 2), 1, max(ny // 8, 2), nx + 11 * ny)
    src = _sources(occ, 3, nx)[:6]
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("kernel", kernel)
    stride = nx * ny + pad
    c.set_option("field_stride", stride)
    for tdt, dt in ((torch.float64, vhp.F64), (torch.float32, vhp.F32)):
        buf = torch.full((len(src) * stride,), -7.0, dtype=tdt, device="cuda")
        d_src = torch.from_numpy(src).cuda()
        c.sweep_batch_device(d_src.data_ptr(), len(src), buf.data_ptr(), dtype=dt)
        c.sync()
        assert c.last_sweep_kernel() == kernel
        got = buf.cpu().numpy().reshape(len(src), stride)
        assert (got[:, nx * ny:] == -7.0).all(), "padding written"
        for k, (sx, sy) in enumerate(src):
            want = oracle.sweep_full(occ, int(sx), int(sy)).astype(got.dtype)
            _assert_same(got[k, :nx * ny].reshape(ny, nx), want, "%dx%d stride +%d kernel %d, source (%d,%d)" % (nx, ny, pad, kernel, sx, sy))


@pytest.mark.parametrize("nx,ny,n", [(1002, 700, 160), (1001, 971, 136), (690, 402, 192), (500, 500, 256)])
def test_pool_is_the_default_for_batches_on_other_widths(vhp, oracle, nx, ny, n):
    # the library's own choice (vhp_capi.hip use_pool_kernel) on widths that are not a multiple of 8: the pool sweep's ANYW build from
    # these batch sizes up; every 8th field against the oracle, all of them written
    import torch
    occ = maps.random_rect_map(nx, ny, 40, 3, nx // 8, 3, ny // 8, nx + 5 * ny)
    src = maps.free_sources(occ, n, 11)
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
    d_out = torch.full((n, ny, nx), float("nan"), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), n, d_out.data_ptr())
    c.sync()
    assert c.last_sweep_kernel() == 3
    assert not bool(torch.isnan(d_out).any())
    for k in range(0, n, 8):
        _assert_same(d_out[k].cpu().numpy(), oracle.sweep_full(occ, int(src[k][0]), int(src[k][1])), "%dx%d default kernel, source %d" % (nx, ny, k))
    c.set_option("kernel", 1)   # ... and the front sweep leaves the same bytes
    d_ref = torch.empty_like(d_out)
    c.sweep_batch_device(d_src.data_ptr(), n, d_ref.data_ptr())
    c.sync()
    assert c.last_sweep_kernel() == 1 and torch.equal(d_ref, d_out)


@pytest.mark.parametrize("kernel", [1, 3, 4])
def test_output_anywhere_on_the_element_grid(vhp, oracle, kernel):
    # the fields of a device batch may start at any element of the caller's buffer (one element in: off the 16-byte grid, off
    # the lines), but not inside an element
    import torch
    nx, ny = 208, 131
    occ = maps.random_rect_map(nx, ny, 20, 1, nx // 8, 1, ny // 8, 99)
    src = _sources(occ, 3, 5)[:6]
    c = vhp.Context(0)
    c.set_map(occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("kernel", kernel)
    d_src = torch.from_numpy(src).cuda()
    for tdt, dt, item in ((torch.float64, vhp.F64, 8), (torch.float32, vhp.F32, 4)):
        for lead in (1, 3, 16):
            buf = torch.full((len(src) * nx * ny + 64,), -7.0, dtype=tdt, device="cuda")
            c.sweep_batch_device(d_src.data_ptr(), len(src), buf.data_ptr() + lead * item, dtype=dt)
            c.sync()
            assert c.last_sweep_kernel() == kernel
            got = buf.cpu().numpy()
            assert (got[:lead] == -7.0).all() and (got[lead + len(src) * nx * ny:] == -7.0).all(), "a store outside the fields"
            got = got[lead: lead + len(src) * nx * ny].reshape(len(src), ny, nx)
            for k, (sx, sy) in enumerate(src):
                _assert_same(got[k], oracle.sweep_full(occ, int(sx), int(sy)).astype(got.dtype), "lead %d kernel %d source (%d,%d)" % (lead, kernel, sx, sy))
        with pytest.raises(vhp.VhpError) as e:
            c.sweep_batch_device(d_src.data_ptr(), len(src), buf.data_ptr() + item // 2, dtype=dt)
        assert e.value.code == vhp.VHP_ERR_ARG


def test_alloc_output_places_a_result_buffer(vhp, oracle):
    # vhp_alloc_output: a result buffer placed by the library (the best of up to N probed allocations); sweeping into it leaves the
    # bytes the same launch leaves anywhere else; freeing is checked
    import torch
    occ, src = maps.config_c3(40)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    n_bytes = 40 * 1000 * 1000 * 8
    ptr, whole, split, tried = c.alloc_output(n_bytes, 6)
    assert ptr and ptr % 256 == 0 and 1 <= tried <= 6
    assert 0.3 < whole < 8.0 and 0.3 < split < 8.0, (whole, split)
    d_src = torch.from_numpy(src).cuda()
    ref = torch.empty((40, 1000, 1000), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), 40, ref.data_ptr())
    c.sweep_batch_device(d_src.data_ptr(), 40, ptr)
    c.sync()
    import ctypes
    got = torch.empty_like(ref)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(got.data_ptr(), ptr, n_bytes, 3) == 0   # device to device
    assert torch.equal(got, ref)
    _assert_same(got[7].cpu().numpy(), oracle.sweep_full(occ, int(src[7][0]), int(src[7][1])), "placed buffer, source 7")
    ms, peak = c.alloc_output_cost()   # what the search cost: wall time, and the memory its candidates held at the peak
    assert ms > 0 and peak == tried * n_bytes
    small, w0, s0, t0 = c.alloc_output(1 << 20, 8)   # below 128 MB nothing can be probed: one allocation, no rates
    assert small and t0 == 1 and w0 == 0.0 and s0 == 0.0
    c.free_output(small)
    c.free_output(ptr)
    with pytest.raises(vhp.VhpError):
        c.free_output(ptr)
    with pytest.raises(vhp.VhpError):
        c.alloc_output(0, 4)


def test_all_256_fields_of_the_bench_launch_on_a_placed_buffer(vhp, oracle):
    # bench.py's second timed region: the C3 launch into a result buffer from vhp_alloc_output.  All 256 fields equal, byte for byte,
    # the same launch into an ordinary allocation -- which test_pool_kernel_all_256_fields_of_the_bench_launch holds against the oracle
    # cell by cell -- and 16 of them are held against the oracle here as well; the search stays inside its budget.
    import ctypes
    import torch
    occ, src = maps.config_c3(256)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    n_bytes = 256 * 1000 * 1000 * 8
    free_before = torch.cuda.mem_get_info()[0]
    ptr, whole, split, tried = c.alloc_output(n_bytes, 16)
    ms, peak = c.alloc_output_cost()
    assert 1 <= tried <= 16 and peak == tried * n_bytes and peak <= free_before // 4 + n_bytes, (tried, peak, free_before)
    d_src = torch.from_numpy(src).cuda()
    ref = torch.full((256, 1000, 1000), float("nan"), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(d_src.data_ptr(), 256, ref.data_ptr())
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemset(ptr, 0xFF, n_bytes) == 0          # (NaN everywhere: a cell the launch does not write would show)
    c.sweep_batch_device(d_src.data_ptr(), 256, ptr)
    c.sync()
    assert c.last_sweep_kernel() == 3
    got = torch.empty_like(ref)
    assert hip.hipMemcpy(got.data_ptr(), ptr, n_bytes, 3) == 0   # device to device
    assert not bool(torch.isnan(got).any())
    assert torch.equal(got, ref)
    for k in range(0, 256, 16):
        _assert_same(got[k].cpu().numpy(), oracle.sweep_full(occ, int(src[k][0]), int(src[k][1])), "placed buffer, source %d" % k)
    c.free_output(ptr)


def test_static_first_round_and_pulled_units_leave_the_same_bytes(vhp, oracle):
    # "pool_static_round": the first unit of every context by workgroup index (the default), or every unit pulled from the queue --
    # the schedule differs, the fields do not; also with fewer units than contexts (the static round is then off by itself)
    import torch
    occ, src = maps.config_c3(128)
    c = _ctx(vhp, occ)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("kernel", 3)
    d_src = torch.from_numpy(src).cuda()
    outs = []
    for static in (2, 0, 1):
        c.set_option("pool_static_round", static)
        for n in (128, 40):
            o = torch.full((n, 1000, 1000), float("nan"), dtype=torch.float64, device="cuda")
            c.sweep_batch_device(d_src.data_ptr(), n, o.data_ptr())
            c.sync()
            assert c.last_sweep_kernel() == 3 and not bool(torch.isnan(o).any())
            outs.append(o)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3]) and torch.equal(outs[0], outs[4]) and torch.equal(outs[1], outs[5])
    assert torch.equal(outs[0][:40], outs[1])
    for k in (0, 39, 127):
        _assert_same(outs[0][k].cpu().numpy(), oracle.sweep_full(occ, int(src[k][0]), int(src[k][1])), "static round, source %d" % k)


def test_field_stride_is_a_property_of_device_buffers(vhp, oracle):
    # "field_stride" places the fields of a DEVICE batch; the host-buffer entry point copies packed fields out of the library's own
    # scratch and ignores it; a stride smaller than a field is refused; vhp_set_map resets it
    import torch
    nx, ny = 200, 163
    occ = maps.random_rect_map(nx, ny, 20, 1, nx // 8, 1, ny // 8, 7)
    src = _sources(occ, 3, 5)[:6]
    c = vhp.Context(0)
    c.set_map(occ)
    want = [oracle.sweep_full(occ, int(sx), int(sy)) for sx, sy in src]
    c.set_option("field_stride", nx * ny + 24)
    got = c.sweep_batch(src)                      # host form: packed, whatever the option says
    for k in range(len(src)):
        _assert_same(got[k], want[k], "host form with field_stride set, source %d" % k)
    c.set_option("field_stride", nx * ny - 8)     # the fields would overlap
    buf = torch.zeros(len(src) * nx * ny, dtype=torch.float64, device="cuda")
    d_src = torch.from_numpy(src).cuda()
    with pytest.raises(vhp.VhpError) as e:
        c.sweep_batch_device(d_src.data_ptr(), len(src), buf.data_ptr())
    assert e.value.code == vhp.VHP_ERR_ARG
    got = c.sweep_batch(src)                      # ... which the host form never sees
    _assert_same(got[0], want[0], "host form with a small field_stride set")
    c.set_option("field_stride", nx * ny + 24)
    occ2 = maps.random_rect_map(nx + 8, ny + 5, 20, 1, nx // 8, 1, ny // 8, 8)   # a larger grid: the old stride would overlap its fields
    c.set_map(occ2)
    src2 = _sources(occ2, 2, 6)[:3]
    buf2 = torch.full((len(src2) * (nx + 8) * (ny + 5),), float("nan"), dtype=torch.float64, device="cuda")
    c.sweep_batch_device(torch.from_numpy(src2).cuda().data_ptr(), len(src2), buf2.data_ptr())
    c.sync()
    got2 = buf2.cpu().numpy().reshape(len(src2), ny + 5, nx + 8)
    for k, (sx, sy) in enumerate(src2):
        _assert_same(got2[k], oracle.sweep_full(occ2, int(sx), int(sy)), "packed again after set_map, source %d" % k)
