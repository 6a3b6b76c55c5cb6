"""The multi-device entry of the C ABI (vhp_multi_*, csrc/vhp_multi.hip): sources block-partitioned over the listed devices,
every device's shard swept on its own stream, fields all-gathered by peer copies.  One GPU here: the ordinal is listed twice
(two contexts, two streams on one device), which walks the same code as two devices; the shard arithmetic is checked against
dist.shard_bounds on the CPU."""
import ctypes as C

import numpy as np
import pytest

import maps


@pytest.fixture(scope="module")
def vhp():
    import torch  # noqa: F401
    import vhp_amd
    return vhp_amd


def test_shard_bounds_match_dist_py():
    from importlib import import_module
    vhp = import_module("visibility-heuristic-path-planner_amd")
    vdist = import_module("visibility-heuristic-path-planner_amd.dist")
    lib = vhp.load_library()
    lo, hi = C.c_int(), C.c_int()
    for n in (0, 1, 5, 8, 255, 256, 1024, 1027):
        for world in (1, 2, 3, 8):
            covered = 0
            for d in range(world):
                lib.vhp_multi_shard_bounds(n, world, d, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == tuple(vdist.shard_bounds(n, d, world)), (n, world, d)
                assert lo.value == covered
                covered = hi.value
            assert covered == n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n_src,n_dev", [(11, 2), (2, 3), (40, 2)])
def test_multi_sweep_and_allgather_on_one_gpu(vhp, oracle, dtype, n_src, n_dev):
    import torch
    lib = vhp.load_library()
    occ = maps.random_rect_map(328, 300, 30, 3, 40, 3, 40, 77)
    src = maps.free_sources(occ, n_src, 5)
    ords = (C.c_int * n_dev)(*([0] * n_dev))
    m = C.c_void_p()
    assert lib.vhp_multi_create(ords, n_dev, C.byref(m)) == 0
    try:
        assert lib.vhp_multi_devices(m) == n_dev
        occ_c = np.ascontiguousarray(occ, np.uint8)
        assert lib.vhp_multi_set_map(m, occ_c.ctypes.data, occ.shape[1], occ.shape[0]) == 0, lib.vhp_multi_last_error(m)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        vdt = vhp.F64 if dtype == "f64" else vhp.F32
        lo, hi = C.c_int(), C.c_int()
        shards, alls, bounds = [], [], []
        for d in range(n_dev):
            lib.vhp_multi_shard_bounds(n_src, n_dev, d, C.byref(lo), C.byref(hi))
            bounds.append((lo.value, hi.value))
            shards.append(torch.full((max(hi.value - lo.value, 1),) + occ.shape, float("nan"), dtype=tdt, device="cuda"))
            alls.append(torch.full((n_src,) + occ.shape, float("nan"), dtype=tdt, device="cuda"))
        torch.cuda.synchronize()
        p_sh = (C.c_void_p * n_dev)(*[t.data_ptr() for t in shards])
        p_all = (C.c_void_p * n_dev)(*[t.data_ptr() for t in alls])
        src_c = np.ascontiguousarray(src, np.int32)
        assert lib.vhp_multi_sweep_batch(m, src_c.ctypes.data, n_src, vhp.SWEEP_FULL, vdt, p_sh) == 0, lib.vhp_multi_last_error(m)
        assert lib.vhp_multi_allgather_fields(m, n_src, vdt, p_sh, p_all) == 0, lib.vhp_multi_last_error(m)
        for k, (sx, sy) in enumerate(src):
            want = oracle.sweep_full(occ, int(sx), int(sy))
            want = want if dtype == "f64" else want.astype(np.float32)
            d = next(i for i, (a, b) in enumerate(bounds) if a <= k < b)
            assert shards[d][k - bounds[d][0]].cpu().numpy().tobytes() == want.tobytes(), "source %d in the shard of device %d" % (k, d)
            for i in range(n_dev):
                assert alls[i][k].cpu().numpy().tobytes() == want.tobytes(), "source %d gathered on device %d" % (k, i)
        # a source outside the grid is reported by the device that swept it
        bad = src_c.copy()
        bad[n_src - 1, 0] = 5000
        assert lib.vhp_multi_sweep_batch(m, bad.ctypes.data, n_src, vhp.SWEEP_FULL, vdt, p_sh) == vhp.VHP_ERR_SOURCE_OOB
    finally:
        assert lib.vhp_multi_destroy(m) == 0
