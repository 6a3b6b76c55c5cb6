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


def test_allgather_plan_gives_every_pair_a_lane_of_its_own():
    # vhp_multi_allgather_plan: the enqueue plan of the peer-copy all-gather as data (host arithmetic: no device needed).  Every
    # (destination, source) pair exactly once; the inbound copies of a destination all on different lanes (= streams: they overlap
    # across its links); in every round (lane k) the sources are a permutation of the devices (no two destinations pull from one
    # source at once); round 0 is the local copy; the pieces are the shards.
    from importlib import import_module
    vhp = import_module("visibility-heuristic-path-planner_amd")
    lib = vhp.load_library()
    for n_src in (0, 1, 5, 8, 255, 1024, 1027):
        for nd in (1, 2, 3, 8):
            cap = nd * nd
            arr = [(C.c_int * cap)() for _ in range(5)]
            n = lib.vhp_multi_allgather_plan(n_src, nd, *[C.cast(a, C.c_void_p) for a in arr], cap)
            to, frm, lane, lo, hi = [list(a[:n]) for a in arr]
            nonempty = [d for d in range(nd) if vhp_shard(lib, n_src, nd, d)[0] < vhp_shard(lib, n_src, nd, d)[1]]
            assert n == nd * len(nonempty)
            assert sorted(zip(to, frm)) == sorted((t, f) for t in range(nd) for f in nonempty)
            for t in range(nd):
                lanes_t = [lane[k] for k in range(n) if to[k] == t]
                assert len(set(lanes_t)) == len(lanes_t), "two inbound copies of device %d share a stream" % t
            for k_round in range(nd):
                srcs = [frm[k] for k in range(n) if lane[k] == k_round]
                assert len(set(srcs)) == len(srcs), "two destinations pull from one source in round %d" % k_round
            for k in range(n):
                assert lane[k] == (frm[k] - to[k]) % nd
                assert (lo[k], hi[k]) == vhp_shard(lib, n_src, nd, frm[k])
            assert lane[:len([k for k in range(n) if lane[k] == 0])] == [0] * len([k for k in range(n) if lane[k] == 0])  # round 0 first
            # a smaller cap truncates the arrays, not the count
            assert lib.vhp_multi_allgather_plan(n_src, nd, None, None, None, None, None, 0) == n
    lo, hi = C.c_int(7), C.c_int(7)
    lib.vhp_multi_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi))   # (no devices: an empty shard, not a division by zero)
    assert (lo.value, hi.value) == (0, 0)


def vhp_shard(lib, n_src, nd, d):
    lo, hi = C.c_int(), C.c_int()
    lib.vhp_multi_shard_bounds(n_src, nd, d, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


@pytest.mark.gpu
def test_multi_rccl_allgather_one_device_and_argument_errors(vhp, oracle):
    # the RCCL form of the all-gather (vhp_multi_use_rccl): a communicator of the one device this box has; a duplicate ordinal is
    # refused; a bad dtype and a missing buffer are refused before anything is enqueued
    import torch
    lib = vhp.load_library()
    occ = maps.random_rect_map(200, 163, 12, 3, 30, 3, 30, 5)
    src = np.ascontiguousarray(maps.free_sources(occ, 6, 3), np.int32)
    occ_c = np.ascontiguousarray(occ, np.uint8)
    m = C.c_void_p()
    assert lib.vhp_multi_create((C.c_int * 2)(0, 0), 2, C.byref(m)) == 0
    try:
        assert lib.vhp_multi_use_rccl(m, 1) == vhp.VHP_ERR_ARG and b"distinct" in lib.vhp_multi_last_error(m)
        assert lib.vhp_multi_set_map(m, occ_c.ctypes.data, occ.shape[1], occ.shape[0]) == 0
        sh = [torch.empty((3,) + occ.shape, dtype=torch.float64, device="cuda") for _ in range(2)]
        p_sh = (C.c_void_p * 2)(*[t.data_ptr() for t in sh])
        assert lib.vhp_multi_sweep_batch(m, src.ctypes.data, 6, vhp.SWEEP_FULL, 77, p_sh) == vhp.VHP_ERR_ARG
        assert lib.vhp_multi_sweep_batch(m, src.ctypes.data, 6, vhp.SWEEP_FULL, vhp.F64, (C.c_void_p * 2)(sh[0].data_ptr(), None)) == vhp.VHP_ERR_ARG
        assert lib.vhp_multi_allgather_fields(m, 6, 77, p_sh, p_sh) == vhp.VHP_ERR_ARG
    finally:
        assert lib.vhp_multi_destroy(m) == 0
    m = C.c_void_p()
    assert lib.vhp_multi_create((C.c_int * 1)(0), 1, C.byref(m)) == 0
    try:
        assert lib.vhp_multi_set_map(m, occ_c.ctypes.data, occ.shape[1], occ.shape[0]) == 0
        rc = lib.vhp_multi_use_rccl(m, 1)
        if rc != 0:
            pytest.skip("librccl not usable here: %r" % lib.vhp_multi_last_error(m))
        shard = torch.full((6,) + occ.shape, float("nan"), dtype=torch.float64, device="cuda")
        allf = torch.full((6,) + occ.shape, float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        p1, p2 = (C.c_void_p * 1)(shard.data_ptr()), (C.c_void_p * 1)(allf.data_ptr())
        assert lib.vhp_multi_sweep_batch(m, src.ctypes.data, 6, vhp.SWEEP_FULL, vhp.F64, p1) == 0, lib.vhp_multi_last_error(m)
        assert lib.vhp_multi_allgather_fields(m, 6, vhp.F64, p1, p2) == 0, lib.vhp_multi_last_error(m)
        for k, (sx, sy) in enumerate(src):
            assert allf[k].cpu().numpy().tobytes() == oracle.sweep_full(occ, int(sx), int(sy)).tobytes()
        assert lib.vhp_multi_use_rccl(m, 0) == 0
    finally:
        assert lib.vhp_multi_destroy(m) == 0
