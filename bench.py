#!/usr/bin/env python3
"""bench.py -- visibility fields/s of the HIP batch sweep (vhp_sweep_batch_device) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one launch of the batched sweep (computeVisibility for S independent
sources) over synthetic input already resident in HBM.  Default workload = BASELINE
config 3, the configuration the metric "visibility fields/sec on 1000x1000 grid at
1/2/4/8 GPUs" is quoted on: 1000x1000 random grid (50 rectangles), 256 seeded sources
per GPU, fp64 fields.  Sources are sharded over ranks (each rank sweeps its own 256;
the sweep has no exchange step, so there is no data-path collective in the timed
region: weak scaling).  `--workload c2` times the single-source README case instead,
`--gather` adds the RCCL all-gather of the per-source fields after each step.  At N > 1 the line also carries
`value_with_collective`: the same job with the fp32 all-gather of the fields and with the max-union + arg-source
(dist.union_fields) after every step, each timed over a few steps behind the main region (compute-only and
compute + collective side by side).

Two numbers, both at the top level of the line, because the memory behind an allocation is of a faster or a slower kind (DESIGN.md
section 7) and the same launch takes 15-20 % longer on the one than on the other:
  * `value` / `roofline.frac` -- the timed region writes into the FIRST allocation of the process, whatever memory it landed on
    (what every round's driver record has measured; also `value_first_allocation`, `roofline.frac_first_allocation`);
  * `value_placed_buffer` / `roofline.frac_placed_buffer` -- the same K steps, same barriers, behind the main region, into a result
    buffer placed by the library's own allocator (vhp_alloc_output: a bounded number of allocations probed, the best kept; what
    the search cost is in `config.output_placement.search_ms` / `.search_bytes_peak`).
`--output-buffer placed` swaps the roles (the main region on the placed buffer).  Where the main region is short (the driver's 20
steps are 10 ms of work) `value_200_steps` / `roofline.kernel_ms_200_steps` repeat it over 200 steps on the same buffer.  `config.output_placement` also says what kind of
memory both buffers and three more allocations of the process are on (vhp_probe_stores), with the launch timed on each.

The defaults (100 steps after 10 warm-up launches, ~0.1 s of GPU time) report the sustained rate.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_CELL = {"f64": 9, "f32": 5}  # 1 B occupancy read + sizeof(field element) written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5", "c1-batch", "c-250", "c3-8", "c3-32", "c3-64", "c3-96", "c1k-empty", "c3-1024", "c3-1016", "c3-512"],
                    help="c3 (default): BASELINE config 3, the configuration the metric is quoted on; c2 / c4 / c5: configs 2, 4, 5 (per-GPU share); "
                         "c1-batch: the reference's own headline size -- config 1's 101x101 rnd_1 mask -- as a batch of 4096 sources; c-250: a 250x250 "
                         "random grid, 256 sources (small-grid batches); c3-8 / c3-32 / c3-64 / c3-96: config 3's map with that many sources -- the batches a planner "
                         "makes of start, goal and frontier pivots; the others are diagnostics")
    ap.add_argument("--sources", type=int, default=0, help="sources per GPU (default: workload's)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--gather", action="store_true", help="all-gather the fields over RCCL inside the timed region")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank sweeps the workload's sources-per-GPU (no collective: the default, BASELINE's metric); "
                         "strong: --sources-total sources are block-partitioned over the ranks (SURVEY 8e: 1024 at config 5)")
    ap.add_argument("--sources-total", type=int, default=1024, help="strong scaling: sources of the whole job")
    ap.add_argument("--gather-mode", default="blocking", choices=["blocking", "overlapped"],
                    help="with --gather: one all-gather after the sweeps, or chunked all-gathers overlapped with the next chunk's sweeps")
    ap.add_argument("--chunks", type=int, default=4, help="pieces per shard for --gather-mode overlapped")
    ap.add_argument("--placements", type=int, default=1,
                    help="diagnostic: allocate this many candidate outputs and time each with a few launches before the timed "
                         "region (a round-1 diagnostic; the probe times are only reported, in config.output_placement)")
    ap.add_argument("--output-buffer", default="first", choices=["placed", "first"],
                    help="where the launches of the MAIN timed region (`value`) write: 'first' = the first allocation of the process, "
                         "whatever memory it landed on; 'placed' = a result buffer from the library's own allocator (vhp_alloc_output: "
                         "up to 16 allocations / 32 GB probed before the timed region, the best kept -- what a caller of the C ABI gets "
                         "who allocates his fields through it).  Either way the OTHER buffer is timed over the same number of steps "
                         "behind the main region and both are reported at the top level (value_first_allocation, value_placed_buffer)")
    ap.add_argument("--kernel", type=int, default=0, choices=[0, 1, 3, 4],
                    help="0 = the library's own choice, 1 = front sweep, 3 = pool sweep, 4 = latency sweep (vhp_set_option \"kernel\")")
    ap.add_argument("--pool-contexts", type=int, default=0, help="pool sweep: units a workgroup holds at once (0 = automatic)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (gloo only with --dry-run or --share-device)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST ONLY: every rank on cuda:0 with the gloo backend, so that the N > 1 code path (broadcast map, barriers, "
                         "max over ranks, the collectives behind the region) runs for real on a one-GPU box; never a result")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the launcher, the rendezvous, the barriers and the max-over-ranks timing with a CPU stub in "
                         "place of the sweep (no GPU, no library): for the CPU test of `bench.py --gpus N`; never a result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_workload(name, rank, n_src):
    import numpy as np
    import vhp_amd
    from importlib import import_module
    synth = import_module("visibility-heuristic-path-planner_amd.synth")
    if name in ("c3-8", "c3-32", "c3-64", "c3-96"):   # config 3's map, the first n of its seeded sources
        n_src = n_src or int(name.split("-")[1])
        name = "c3"
    if name == "c3":
        n = n_src or 256
        # (rank 0's launch IS synth.config_c3(n): the launch tests/test_gpu_sweep.py::test_config3_all_256_fields... checks field by
        # field against the oracle; tests/test_bench_launcher.py keeps the two recipes in lock-step)
        occ, src = synth.config_c3(n)
        if rank:
            src = synth.free_sources(occ, n, seed=7 + 1000 * rank)
        label = "C3: 1000x1000 random grid (50 rectangles 20..100, map seed 1), %d seeded sources per GPU" % n
    elif name in ("c3-1024", "c3-1016", "c3-512"):  # diagnostic: C3 on a pitch that is a multiple of 128 B / of 64 B only
        side = int(name.split("-")[1])
        n = n_src or 256
        occ = synth.random_rect_map(side, side, 50, 20, 100, 20, 100, seed=1)
        src = synth.free_sources(occ, n, seed=7 + 1000 * rank)
        label = "C3 variant: %dx%d random grid, %d seeded sources per GPU" % (side, side, n)
    elif name == "c5":
        n = n_src or 128
        occ, src = synth.config_c5(n)
        if rank:
            src = synth.free_sources(occ, n, seed=11 + 1000 * rank)
        label = "C5: 4096x4096 random grid (50 rectangles 80..400, map seed 1), %d seeded sources per GPU" % n
    elif name == "c1-batch":
        n = n_src or 4096
        occ = synth.c1_rnd1_mask()
        src = synth.free_sources(occ, n, seed=7 + 1000 * rank)   # (6 909 free cells; a source may repeat: every field is swept all the same)
        label = "C1 as a batch: 101x101 mask from rnd_1.mat (the reference's headline size), %d seeded sources per GPU" % n
    elif name == "c-250":
        n = n_src or 256
        occ = synth.random_rect_map(250, 250, 50, 5, 25, 5, 25, seed=1)
        src = synth.free_sources(occ, n, seed=7 + 1000 * rank)
        label = "250x250 random grid (50 rectangles 5..25, map seed 1), %d seeded sources per GPU" % n
    elif name == "c2":
        n = n_src or 1
        occ = np.ones((1000, 1000), np.uint8)
        src = np.array([[500, 500]] * n, np.int32)
        label = "C2: 1000x1000 empty grid, centre source, %d source(s) per launch" % n
    else:
        n = n_src or 256
        occ = np.ones((1000, 1000), np.uint8)
        src = synth.free_sources(occ, n, seed=7 + 1000 * rank)
        label = "1000x1000 empty grid, %d seeded sources per GPU" % n
    return occ, src, label


def cpu_baseline(occ, src, seconds):
    """Times the CPU oracle (the port of computeVisibility) on this host's cores.

    Built here, on the machine that runs it, with the flags the reference ships
    (-O3 -Ofast -march=native, CMakeLists.txt:10); falls back to the strict build.
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    fast = os.path.join(ROOT, "oracle", "libvhp_oracle_fast.so")
    flags = "-O3 -Ofast -march=native"
    fallback = ""
    try:
        if os.path.exists(fast):
            os.remove(fast)
        oracle_lib.build(fast=True)
        orc = oracle_lib.Oracle(fast)
    except (OSError, AttributeError, subprocess.CalledProcessError) as e:
        # the reference-flags build could not be made or loaded on this host: time the strict build and SAY SO
        oracle_lib.build()
        orc = oracle_lib.Oracle()
        flags = "-O2 -ffp-contract=off"
        fallback = "; FALLBACK to the strict build because the -Ofast build failed: %r" % (e,)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # single thread (the reference's mode)
    n1 = max(4, min(len(src), 16))
    wall1, best1 = orc.time_sweeps(occ, src[:n1], 1)
    one = n1 / wall1
    # all cores, one independent source per thread, bounded to ~`seconds`
    per = max(1, min(len(src), cores * 2))
    done, t0 = 0, time.time()
    wall = 0.0
    while True:
        w, _ = orc.time_sweeps(occ, src[:per] if per <= len(src) else src, cores)
        wall += w
        done += min(per, len(src))
        if time.time() - t0 > seconds:
            break
    return {
        "value": round(done / wall, 2), "unit": "fields/s", "cores": cores, "kind": "port",
        "flags": flags, "fallback": bool(fallback),
        "sample": "oracle computeVisibility port (%s), %d sweeps of the same %dx%d workload on %d threads; "
                  "single thread: %.1f fields/s (best sweep %.2f ms)%s" % (flags, done, occ.shape[1], occ.shape[0], cores, one, best1 * 1e3, fallback),
        "single_thread_value": round(one, 2),
    }


def bench_planner(args):
    """--workload c4: BASELINE config 4, the full visibility-heuristic planner on maze_6 (690x402, threshold 0.1, 64 pivots,
    results left on the device).  A step = one whole solve.  The planner does not shard (pivot k+1 needs the union after
    pivot k): one GPU; with --gpus N every rank solves its own replica."""
    import numpy as np
    import torch
    import vhp_amd
    from importlib import import_module
    synth = import_module("visibility-heuristic-path-planner_amd.synth")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    occ = synth.maze_6()
    ny, nx = occ.shape
    start, end = (345, ny - 1 - 391), (341, ny - 1 - 10)
    ctx = vhp_amd.Context(local_rank)
    ctx.set_map(occ)
    n_piv = 0
    for _ in range(max(args.warmup, 1)):
        rc, n_piv, _ = ctx.planner_solve_device(start, end, 0.1, 250)
        assert rc == 0
    torch.cuda.synchronize()
    dev_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.planner_solve_device(start, end, 0.1, 250)
        dev_ms.append(ctx.last_elapsed_ms())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loop_ms = float(np.mean(dev_ms))
    bytes_per_pivot = 32 * nx * ny  # vis_local 8 W + 8 R, vis_global 8 R + 8 W, labels 4 R + 4 W
    achieved = bytes_per_pivot * n_piv / (loop_ms * 1e-3) / 1e9
    out = {
        "metric": "planner pivots/sec on maze_6 (690x402)", "value": round(n_piv * args.steps / elapsed, 1), "unit": "pivots/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "reference fixture (images/maze_6.png as tests/golden/maze_6.npz)",
        "config": {"workload": "C4: maze_6, start {345,391}, end {341,10}, visibilityThreshold 0.1: %d pivots per solve, results device-resident" % n_piv,
                   "us_per_pivot_device_loop": round(loop_ms * 1e3 / n_piv, 2), "us_per_pivot_wall": round(elapsed / args.steps * 1e6 / n_piv, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": None, "kernel": "sweep of the iteration (vhp_lat_sweep where a batch of one takes it, else vhp_planner_sweep) + vhp_planner_epilogue", "kernel_ms": round(loop_ms, 4),
                     "note": "latency case by construction: one source per pivot, pivots are sequential (SURVEY 8d expectation management)"},
    }
    # the speculative planner (vhp_planner_solve_speculative): exact mode (same pivots, cached fields) and fast mode (all k
    # candidates of a launch committed: NOT the reference's result), device loop only; "vs_plain_loop" = the plain loop's time / theirs
    spec = {}
    for name, k, mode in (("exact_k2", 2, 0), ("exact_k4", 4, 0), ("fast_k2", 2, 1), ("fast_k4", 4, 1), ("fast_k8", 8, 1)):
        r = None
        ms = []
        for it in range(1 + min(args.steps, 5)):
            r = ctx.planner_solve_speculative(start, end, 0.1, 400, k=k, mode=mode, outputs=False)
            if it:
                ms.append(ctx.last_elapsed_ms())
        spec[name] = {"status": r["status"], "pivots": r["n_pivots"], "sweep_launches": r["sweeps"], "cache_hits": r["hits"], "fields_swept": r["fields_swept"],
                      "device_loop_ms": round(float(np.mean(ms)), 4), "us_per_pivot_device_loop": round(float(np.mean(ms)) * 1e3 / max(r["n_pivots"], 1), 2),
                      "us_per_launch": round(float(np.mean(ms)) * 1e3 / max(r["sweeps"] + r["hits"], 1), 2),
                      "vs_plain_loop": round(loop_ms / float(np.mean(ms)), 3)}   # > 1: faster than vhp_planner_solve's device loop
    out["config"]["speculative"] = spec
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        # the port of solve() built with the flags the reference ships (-O3 -Ofast -march=native), the strict build if that fails
        flags, fallback = "-O3 -Ofast -march=native", ""
        try:
            fast = os.path.join(ROOT, "oracle", "libvhp_oracle_fast.so")
            if os.path.exists(fast):
                os.remove(fast)
            oracle_lib.build(fast=True)
            orc = oracle_lib.Oracle(fast)
        except (OSError, AttributeError, subprocess.CalledProcessError) as e:
            oracle_lib.build()
            orc = oracle_lib.Oracle()
            flags, fallback = "-O2 -ffp-contract=off", "; FALLBACK to the strict build because the -Ofast build failed: %r" % (e,)
        ts = []
        t_end = time.time() + min(args.cpu_seconds, 10.0)
        while time.time() < t_end or len(ts) < 3:
            t = time.perf_counter()
            w = orc.solve(occ, start, end, 0.1, 250)
            ts.append(time.perf_counter() - t)
        out["cpu_baseline"] = {"value": round(w["n_pivots"] / min(ts), 1), "unit": "pivots/s", "cores": 1, "kind": "port", "flags": flags, "fallback": bool(fallback),
                               "sample": "oracle port of solve() (%s), best of %d solves of the same maze (the reference's loop is one thread)%s" % (flags, len(ts), fallback)}
    print(json.dumps(out), flush=True)


def launch_workers(args):
    """`python bench.py --gpus N` typed by hand (no torchrun around it): start the N ranks ourselves.

    Must run before anything in this process touches the GPU (a process that has initialised HIP must not start
    replacing itself, and the children need the devices untouched): it imports nothing but the standard library.  The
    children are `python -m torch.distributed.run` workers of this same file with the same arguments; rank 0's JSON line
    comes through on stdout; the exit code is the launcher's (non-zero if any rank failed)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: the launcher picks a free rendezvous port itself (no bind-close-reuse race on a busy host)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=%d" % args.gpus, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def bench_dry_run(args, rank, world):
    """--dry-run: the protocol of main() -- rendezvous, warm-up, barrier, K timed steps, barrier, max over ranks, one JSON
    line from rank 0 -- with a CPU stub as the step.  Checks plumbing only; prints "dry_run": true and no roofline."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from importlib import import_module
    vdist = import_module("visibility-heuristic-path-planner_amd.dist")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    n_src = args.sources or 256
    lo, hi = vdist.shard_bounds(world * n_src, rank, world)       # this rank's block of the job's sources
    work = np.arange(lo, hi, dtype=np.float64)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        work.sum()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        work.sum()
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    with_coll = None
    if world > 1:  # the same two collectives as the real run, on stand-in fields of 8 x 8 cells
        fields = torch.full((hi - lo, 8, 8), float(rank))
        with_coll = {}
        for name, fn in (("allgather_f32", lambda: vdist.gather_fields(fields.float(), world * n_src)),
                         ("union_fields", lambda: vdist.union_fields(fields, lo, world * n_src))):
            barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                work.sum()
                fn()
            barrier()
            tc = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            with_coll[name] = {"value": round(world * n_src * 3 / float(tc.item()), 2), "unit": "fields/s", "steps": 3}
    if rank == 0:
        print(json.dumps({"metric": "visibility fields/sec on 1000x1000 grid", "value": round(world * n_src * args.steps / float(t.item()), 2),
                          "value_with_collective": with_coll,
                          "unit": "fields/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(t.item()) / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "dry_run": True, "config": {"workload": "DRY RUN: CPU stub, no sweep", "sources_per_gpu": n_src,
                                                       "sharding": "sources/%d" % world, "backend": args.backend}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    if args.dry_run:
        return bench_dry_run(args, rank, world)
    if args.backend != "nccl" and not args.share_device:
        raise SystemExit("bench.py: --backend gloo exists for --dry-run and --share-device only (the sweep runs on the GPU, RCCL is its collective)")
    if args.share_device and args.backend != "gloo":
        raise SystemExit("bench.py: --share-device needs --backend gloo (RCCL refuses two ranks on one device)")
    if args.workload == "c4":
        return bench_planner(args)
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    import vhp_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from importlib import import_module
    vdist = import_module("visibility-heuristic-path-planner_amd.dist")
    if args.scaling == "strong":
        # the whole job's sources, block-partitioned over the ranks (dist.shard_bounds)
        total = args.sources_total
        if args.gather and args.gather_mode == "overlapped":
            total -= total % (world * args.chunks)
        occ, src_all, label = make_workload(args.workload, 0, total)
        lo, hi = vdist.shard_bounds(total, rank, world)
        src = src_all[lo:hi]
        label = label.replace("per GPU", "in total, %d per GPU" % (hi - lo))
    else:
        occ, src, label = make_workload(args.workload, rank, args.sources)
    ny, nx = occ.shape
    n_src = len(src)
    ctx = vhp_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    if world > 1:
        # the occupancy map is broadcast from rank 0 over RCCL and stays on the device
        d_occ = torch.from_numpy(occ).to(dev) if rank == 0 else torch.zeros((ny, nx), dtype=torch.uint8, device=dev)
        vdist.broadcast_map(d_occ, 0)
        torch.cuda.synchronize()
        ctx.set_map_device(d_occ.data_ptr(), nx, ny)
    else:
        ctx.set_map(occ)  # uploads + packs: the map is resident before the timed region
    if args.kernel:
        ctx.set_option("kernel", args.kernel)
    if args.pool_contexts:
        ctx.set_option("pool_contexts", args.pool_contexts)
    d_src = torch.from_numpy(np.ascontiguousarray(src, np.int32)).to(dev)
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    vdt = vhp_amd.F64 if args.dtype == "f64" else vhp_amd.F32
    # The first allocation of the process is made here, before anything else asks for device memory.  (--placements N > 1: a round-1
    # diagnostic that times N candidate allocations first and reports them.)
    out_bytes = n_src * ny * nx * (8 if args.dtype == "f64" else 4)
    n_cand = max(1, min(args.placements, int((24 << 30) // max(out_bytes, 1))))
    cands, probe_ms = [], []
    for _ in range(n_cand):
        cands.append(torch.empty((n_src, ny, nx), dtype=tdt, device=dev))
    if n_cand > 1:
        for buf in cands:
            for _ in range(2):
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, buf.data_ptr(), dtype=vdt)
            torch.cuda.synchronize()
            ctx.timing(True, prealloc=8)
            for _ in range(5):
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, buf.data_ptr(), dtype=vdt)
            torch.cuda.synchronize()
            probe_ms.append(round(float(np.median(ctx.timing_collect(5))), 4))
            ctx.timing(False)
    d_first = cands[0]   # the first allocation of the process
    cands = None
    torch.cuda.empty_cache()
    gathered = None
    overlapped = args.gather and args.gather_mode == "overlapped" and world >= 1 and n_src % args.chunks == 0 and dist.is_initialized()
    # A second result buffer, placed by the library (vhp_alloc_output, include/vhp.h: the memory behind an allocation is of a faster or a
    # slower kind and no allocation API chooses, so the library tries a bounded number of allocations, probes each and keeps the best:
    # DESIGN.md appendix A.7) -- set up here, before any timed region, like the map and the sources.  --output-buffer first (the default):
    # the main region writes into the first allocation and the placed buffer is timed behind it; placed: the other way round.
    placed = None
    d_placed = None
    if not args.gather and out_bytes >= (128 << 20):   # (every rank: the region behind the main one has the same barriers)
        try:
            n_try = max(1, min(16, int((32 << 30) // max(out_bytes, 1))))
            p_ptr, p_w, p_sp, p_tried = ctx.alloc_output(out_bytes, n_try)
            s_ms, s_peak = ctx.alloc_output_cost()

            class _Placed:  # (what the launches need of a tensor)
                def data_ptr(self):
                    return p_ptr
            d_placed = _Placed()
            placed = {"policy": "vhp_alloc_output: up to %d allocations (32 GB) probed, the best kept" % n_try, "allocations_tried": p_tried,
                      "search_ms": round(s_ms, 1), "search_bytes_peak": int(s_peak),
                      "whole_lines_TBps": round(p_w, 2), "split_lines_TBps": round(p_sp, 2)}
        except Exception as e:  # (out of memory on a shared device: the first allocation only, then)
            placed = None
            d_placed = None
            sys.stderr.write("vhp_alloc_output failed (%r): the first allocation only\n" % (e,))
    if world > 1:
        # every rank runs the same regions with the same barriers: a placed buffer on all of them, or on none
        have = torch.tensor([1 if d_placed is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) == 0 and d_placed is not None:
            ctx.free_output(d_placed.data_ptr())
            d_placed, placed = None, None
    d_out = d_placed if (args.output_buffer == "placed" and d_placed is not None) else d_first
    d_other = None if d_placed is None else (d_first if d_out is d_placed else d_placed)
    if args.gather and (world > 1 or overlapped):
        gathered = torch.empty((world * n_src, ny, nx), dtype=tdt, device=dev)

    def launch_piece(a, b, dst):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.sweep_batch_device(d_src[a:b].data_ptr(), b - a, dst.data_ptr(), dtype=vdt)

    def step(dst=None):
        if overlapped:
            vdist.sweep_gather_overlapped(launch_piece, range(world * n_src), gathered, args.chunks)
            return
        ctx.sweep_batch_device(d_src.data_ptr(), n_src, (dst if dst is not None else d_out).data_ptr(), dtype=vdt)
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, d_out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    launches_per_step = args.chunks if overlapped else 1

    def timed_region(dst=None, steps=None):
        """W warm-up steps, barrier, K timed steps, barrier: (seconds -- max over ranks --, this rank's kernel ms per step)."""
        steps = steps or args.steps
        for _ in range(args.warmup):
            step(dst)
        barrier()
        ctx.timing(True, prealloc=steps * launches_per_step + 2)  # HIP events around every sweep kernel, on the stream it is launched on;
        #                                              the event pairs exist before the timed region starts
        t0 = time.perf_counter()
        for _ in range(steps):
            step(dst)
        barrier()
        el = time.perf_counter() - t0
        ctx.sync()  # surfaces device-side validation errors
        k = ctx.timing_collect(steps * launches_per_step)
        ctx.timing(False)
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), float(k.sum()) / steps

    elapsed, kern_ms = timed_region()
    # the same K steps into the OTHER result buffer (the library-placed one, or with --output-buffer placed the first allocation)
    other = None
    if d_other is not None:
        other = timed_region(d_other)
    # ... and, where the main region is short (the driver's 20 steps are 10 ms of work), the same buffer again over 200 steps: the
    # sustained rate beside the line's own (reported, never `value`)
    sustained = None
    if args.steps < 200 and not args.gather:
        sustained = timed_region(None, 200)
    # Not a parity test (tests/test_gpu_*.py hold these very launches against the oracle field by field), a tripwire: the first fields
    # the timed launches left in the first allocation, byte for byte against ANOTHER kernel's -- the front sweep's -- on the same
    # sources.  Behind both timed regions, before anything else writes the buffer.
    self_check = None
    timed_kernel = ctx.last_sweep_kernel()   # what the timed launches ran (the self-check below launches another kernel)
    if rank == 0 and not overlapped:
        try:
            n_chk = min(8, n_src)
            ref = torch.empty((n_chk, ny, nx), dtype=tdt, device=dev)
            ctx.set_option("kernel", 1)
            ctx.sweep_batch_device(d_src.data_ptr(), n_chk, ref.data_ptr(), dtype=vdt)
            ctx.sync()
            ctx.set_option("kernel", args.kernel)
            self_check = {"fields_compared": n_chk, "timed_kernel": timed_kernel, "against_kernel": 1,
                          "equal": bool(torch.equal(ref, d_first[:n_chk]))}
            ref = None
        except Exception as e:
            self_check = {"error": repr(e)}

    # ---- behind the timed region: the job with its collective (N > 1), then what memory the timed buffer was -----------------
    with_coll = None
    if world > 1 and not args.gather and args.workload in ("c3", "c5"):
        # compute-only (`value`) and compute + collective side by side.  (a) the all-gather of per-source fields
        # north_star names, in fp32 (the fp64 result of config 5 alone is 137 GB per GPU); (b) what the planner needs of a
        # sharded batch: the max-union and the arg-source (dist.union_fields).  A few steps each, same barriers, max over ranks.
        with_coll = {}
        k_coll = max(2, min(args.steps, 10))
        try:
            out32 = torch.empty((n_src, ny, nx), dtype=torch.float32, device=dev)
            all32 = torch.empty((world * n_src, ny, nx), dtype=torch.float32, device=dev)
            first = rank * n_src

            def step_gather():
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, out32.data_ptr(), dtype=vhp_amd.F32)
                dist.all_gather_into_tensor(all32, out32)

            def step_union():
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, d_first.data_ptr(), dtype=vdt)
                vdist.union_fields(d_first, first, world * n_src, ctx=ctx)   # (one pass of the HIP max-union kernel per rank, the partials all-gathered)

            for name, fn in (("allgather_f32", step_gather), ("union_fields", step_union)):
                fn()
                barrier()
                t1 = time.perf_counter()
                for _ in range(k_coll):
                    fn()
                barrier()
                tc = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                dist.all_reduce(tc, op=dist.ReduceOp.MAX)
                with_coll[name] = {"value": round(world * n_src * k_coll / float(tc.item()), 2), "unit": "fields/s", "steps": k_coll,
                                   "ms_per_step": round(float(tc.item()) / k_coll * 1e3, 4)}
            out32 = all32 = None
            torch.cuda.empty_cache()
        except Exception as e:  # (memory: config 5 gathered on every rank is 69 GB in fp32)
            with_coll["error"] = repr(e)
    on_placed = d_placed is not None and d_out is d_placed
    placement = {"timed_on": ("library-placed buffer (vhp_alloc_output, set up before the timed region)" if on_placed else "first allocation of the process"),
                 "candidates": n_cand, "probe_kernel_ms": probe_ms,
                 "probe_median_ms": (round(float(np.median(probe_ms)), 4) if probe_ms else None),
                 "probe_best_ms": (min(probe_ms) if probe_ms else None)}
    if placed:
        placement["search_ms"] = placed["search_ms"]
        placement["search_bytes_peak"] = placed["search_bytes_peak"]
    if rank == 0 and out_bytes >= (128 << 20) and not overlapped:
        # What kind of memory are the two buffers on?  Two store patterns (vhp_probe_stores: whole lines / lines written in halves,
        # plain stores) on the first allocation (the placed buffer was probed by the search), and the same plus a few launches of the
        # sweep on three more allocations of this process.
        def state_of(split):
            return "fast" if split >= 4.6 else "slow" if split <= 4.1 else "mixed"

        def launch_ms(buf, k=5):
            for _ in range(2):
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, buf.data_ptr(), dtype=vdt)
            torch.cuda.synchronize()
            ctx.timing(True, prealloc=k + 2)
            for _ in range(k):
                ctx.sweep_batch_device(d_src.data_ptr(), n_src, buf.data_ptr(), dtype=vdt)
            torch.cuda.synchronize()
            v = float(np.median(ctx.timing_collect(k)))
            ctx.timing(False)
            return round(v, 4)

        note = ("vhp_probe_stores: 1 KB row pieces in many streams, plain stores; on this device an allocation answers whole lines with "
                "4.7-5.0 (slow) or 5.6-6.1 TB/s (fast), split lines with 3.5-3.7 or 4.7-5.4, DESIGN.md appendix A.7")
        try:
            w, sp = ctx.probe_stores(d_first.data_ptr(), out_bytes)
            placement["first_allocation_of_this_process"] = {"whole_lines_TBps": round(w, 2), "split_lines_TBps": round(sp, 2), "state": state_of(sp), "note": note}
            if placed:
                placement["library_placed_buffer"] = dict(placed, state=state_of(placed["split_lines_TBps"]))
            others = []
            n_more = max(0, min(3, int((40 << 30) // max(out_bytes, 1)) - 1))
            keep = []
            for _ in range(n_more):
                b = torch.empty((n_src, ny, nx), dtype=tdt, device=dev)
                keep.append(b)
                ms = launch_ms(b)
                w, sp = ctx.probe_stores(b.data_ptr(), out_bytes)
                others.append({"kernel_ms": ms, "whole_lines_TBps": round(w, 2), "split_lines_TBps": round(sp, 2), "state": state_of(sp)})
            placement["other_allocations_of_this_process"] = others
            keep = None
            torch.cuda.empty_cache()
        except Exception as e:
            placement["error"] = repr(e)

    if rank == 0:
        fields = world * n_src * args.steps
        alg_bytes = BYTES_PER_CELL[args.dtype] * nx * ny * n_src  # per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # the main region and the region on the other buffer, by name: (seconds, kernel ms) of each
        first_reg = (elapsed, kern_ms) if not on_placed else other
        placed_reg = (elapsed, kern_ms) if on_placed else other
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of this
        # command): collected by tools/collect_profiles.sh into profiles/, not measured inside this run
        traffic, traffic_src = None, None
        # what the library launched IN THE TIMED REGION (recorded there: the self-check behind it launches the front sweep)
        kname = {1: "vhp_sweep_fronts", 3: "vhp_pool_sweep", 4: "vhp_lat_sweep"}.get(timed_kernel, "unknown")
        tpath = os.path.join(ROOT, "profiles", "traffic_%s_%s_%s.json" % (args.workload, args.dtype, kname))
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_src = "static: %s (rocprofv3 --pmc passes of this command, not measured in this run)" % os.path.relpath(tpath, ROOT)
            except Exception:
                traffic = None
        out = {
            "metric": "visibility fields/sec on %dx%d grid" % (nx, ny),
            "value": round(fields / elapsed, 2),
            "value_first_allocation": (round(fields / first_reg[0], 2) if first_reg else None),
            "value_placed_buffer": (round(fields / placed_reg[0], 2) if placed_reg else None),
            "value_200_steps": (round(world * n_src * 200 / sustained[0], 2) if sustained else None),
            "unit": "fields/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": label, "grid": [nx, ny], "sources_per_gpu": n_src, "sharding": "sources/%d" % world,
                       "kernel_option": args.kernel,
                       "self_check": self_check,
                       "output_placement": placement,
                       "collective": ("rccl all_gather of fields (%s%s)" % (args.gather_mode if gathered is not None else "", ", %d chunks per shard" % args.chunks if overlapped else "")
                                      if gathered is not None else "none (independent sources)"),
                       "map": "rccl broadcast from rank 0" if world > 1 else "uploaded by the only rank",
                       **({"multi_gpu": "the builder has never run this path across devices (one-GPU boxes only): first run on a node is the driver's"} if world > 1 else {}),
                       **({"share_device": "TEST MODE: every rank on cuda:0, gloo collectives -- not a result"} if args.share_device else {})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": round(kern_ms, 4),
                         "frac_first_allocation": (round(alg_bytes / (first_reg[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if first_reg else None),
                         "kernel_ms_first_allocation": (round(first_reg[1], 4) if first_reg else None),
                         "frac_placed_buffer": (round(alg_bytes / (placed_reg[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if placed_reg else None),
                         "kernel_ms_placed_buffer": (round(placed_reg[1], 4) if placed_reg else None),
                         "kernel_ms_200_steps": (round(sustained[1], 4) if sustained else None),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # the same rate counting only the field bytes written (the occupancy maps are read at 2 bits/cell)
                         "frac_field_bytes": round((alg_bytes - nx * ny * n_src) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if with_coll is not None:
            out["value_with_collective"] = with_coll
        if not args.no_cpu_baseline:  # (rank 0, at any N: the other ranks wait at the last barrier)
            try:
                out["cpu_baseline"] = cpu_baseline(occ, src, args.cpu_seconds)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "fields/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
