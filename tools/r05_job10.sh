#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job10; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
for w in c1-batch c-250 c2 c5; do timeout 300 python3 bench.py --workload $w --steps 20 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python3 bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -4 $O/pytest_gpu.log; for w in c1-batch c-250 c2 c5 c4 driver_cmd; do echo "== $w"; cut -c1-330 $O/bench_$w.json; python3 -c "
import json,sys
d=json.load(open('$O/bench_$w.json'))
print('   roofline', {k:d['roofline'].get(k) for k in ('frac','kernel','kernel_ms','frac_first_allocation','frac_placed_buffer')}, 'cpu', d.get('cpu_baseline',{}).get('value'))
"; tail -2 $O/bench_$w.err; done
