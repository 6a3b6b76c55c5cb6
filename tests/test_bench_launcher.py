"""`python bench.py --gpus N` as the driver types it: when no launcher is around it (WORLD_SIZE unset) bench.py starts the N
ranks itself, before anything touches a GPU.  Driven here on CPU with the gloo backend and the --dry-run stub in place of
the sweep: what is checked is the launcher, the rendezvous on 127.0.0.1, the barriers, the max over ranks and that exactly
one JSON line with "n_gpus": N comes out -- and that a failing rank fails the command."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=env)


def _json_lines(out):
    return [json.loads(line) for line in out.splitlines() if line.startswith("{")]


def test_gpus_2_spawns_two_ranks():
    r = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--sources", "100")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["steps"] == 4 and line["warmup"] == 1
    assert line["config"]["sharding"] == "sources/2" and line["scaling"] == "weak"
    # whole-job aggregate: both ranks' sources over the slowest rank's time
    assert abs(line["value"] - 2 * 100 * 4 / (line["ms_per_step"] * 4e-3)) <= 0.02 * line["value"]
    # at N > 1 the line carries the job with its collective next to the compute-only value, for both collectives
    wc = line["value_with_collective"]
    assert set(wc) == {"allgather_f32", "union_fields"}
    for v in wc.values():
        assert v["unit"] == "fields/s" and v["value"] > 0 and v["steps"] == 3


def test_gpus_1_runs_in_process():
    r = _run("--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_lines(r.stdout)[0]
    assert line["n_gpus"] == 1 and line["value_with_collective"] is None   # one rank: nothing to exchange


def test_a_failing_rank_fails_the_command():
    # gloo is refused outside --dry-run (the sweep has no CPU path): every rank exits non-zero and so must the launcher
    r = _run("--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_bench_workloads_are_the_tested_launches():
    # bench.py's timed region checks no results; what makes its numbers results of a CORRECT launch is that rank 0's launch of
    # the default workload is, input for input, the launch tests/test_gpu_sweep.py checks against the oracle (maps.config_c3 /
    # config_c5): same map, same sources, same order
    import importlib.util
    import numpy as np
    import maps
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name, ref in (("c3", maps.config_c3(256)), ("c5", maps.config_c5(128))):
        occ, src, _ = bench.make_workload(name, 0, 0)
        assert occ.shape == ref[0].shape and np.array_equal(occ, ref[0])
        assert np.array_equal(np.asarray(src), np.asarray(ref[1]))
    # other ranks sweep other sources of the same map
    occ1, src1, _ = bench.make_workload("c3", 1, 0)
    assert np.array_equal(occ1, maps.config_c3(1)[0]) and not np.array_equal(np.asarray(src1), np.asarray(maps.config_c3(256)[1]))
