#!/bin/bash
# A/B of the two batch-sweep kernels in fresh processes: a process per run, its first allocation the output (--output-buffer first),
# kernels alternating.  usage: bash tools/ab_bench.sh [runs] [extra bench.py args]
runs=${1:-4}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $runs); do
  for k in 1 3; do
    python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --output-buffer first --kernel $k "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('run $i kernel option $k: %s kernel_ms %.4f ms_per_step %.4f frac %.3f' % (r['kernel'], r['kernel_ms'], d['ms_per_step'], r['frac']))"
  done
done
