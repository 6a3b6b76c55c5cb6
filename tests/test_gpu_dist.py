"""The multi-GPU source sharding with the HIP sweep as the per-rank compute, under torch.distributed's RCCL backend
("nccl") with a single rank: every code path of dist.py that the 8-GPU bench uses (map broadcast, block partition,
all-gather of fields, max-union + arg-source, chunked all-gather overlapped with the next chunk's sweeps) runs on one
MI355X, on device tensors, against the oracle.  World sizes 2 and 3 run on CPU with gloo (tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest

import maps

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_hip_sweep_under_rccl_single_rank(oracle):
    import torch
    import torch.distributed as dist
    import vhp_amd
    from importlib import import_module
    vd = import_module("visibility-heuristic-path-planner_amd.dist")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        occ = maps.random_rect_map(328, 300, 30, 3, 40, 3, 40, 5)
        n = 16
        src = maps.free_sources(occ, n, 8)
        full = np.stack([oracle.sweep_full(occ, int(x), int(y)) for x, y in src])
        # the map is broadcast as a device tensor and handed to the library without a host round trip
        d_occ = vd.broadcast_map(torch.from_numpy(occ).cuda(), 0)
        ctx = vhp_amd.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_map_device(d_occ.data_ptr(), occ.shape[1], occ.shape[0])

        def compute(shard):
            d_src = torch.from_numpy(np.ascontiguousarray(shard, np.int32)).cuda()
            out = torch.empty((len(shard),) + occ.shape, dtype=torch.float64, device="cuda")
            if len(shard):
                ctx.sweep_batch_device(d_src.data_ptr(), len(shard), out.data_ptr())
            ctx.sync()
            return out

        local, lo = vd.sweep_sharded(compute, src, "none")
        assert lo == 0 and np.array_equal(local.cpu().numpy(), full)
        allf = vd.sweep_sharded(compute, src, "gather")
        assert np.array_equal(allf.cpu().numpy(), full)
        best, arg = vd.sweep_sharded(compute, src, "union")
        assert np.array_equal(best.cpu().numpy(), full.max(0))
        assert np.array_equal(arg.cpu().numpy(), (full == full.max(0)[None]).argmax(0))
        # chunked all-gather overlapped with the sweeps of the next chunk, on two streams
        d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).cuda()
        out = torch.full((n,) + occ.shape, -1.0, dtype=torch.float64, device="cuda")

        def launch(a, b, dst):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.sweep_batch_device(d_src[a:b].data_ptr(), b - a, dst.data_ptr())

        vd.sweep_gather_overlapped(launch, src, out, chunks=4)
        torch.cuda.synchronize()
        ctx.sync()
        rows = [vd.gathered_index(i, n, 1, 4) for i in range(n)]
        assert np.array_equal(out.cpu().numpy()[rows], full)
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_on_one_device():
    # bench.py's N > 1 path for real (not --dry-run): two ranks, both on cuda:0 (--share-device: gloo collectives, since RCCL refuses two
    # ranks on one device), the map broadcast from rank 0, both timed regions with their barriers, the collectives behind them, the
    # max over ranks, one JSON line from rank 0.  What the driver's 8-GPU run walks, minus RCCL itself (test above).
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=2",
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3", "--warmup", "1",
           "--sources", "48", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["sources_per_gpu"] == 48 and d["config"]["sharding"] == "sources/2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 48 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]
    assert d["value_first_allocation"] == d["value"] and d["roofline"]["kernel"] in ("vhp_pool_sweep", "vhp_sweep_fronts", "vhp_lat_sweep")
    assert d["config"]["self_check"]["equal"] is True
    wc = d["value_with_collective"]
    assert "error" not in wc and wc["allgather_f32"]["value"] > 0 and wc["union_fields"]["value"] > 0, wc
