#!/bin/bash
# Collects the artifacts that profiles/ holds for a round, in one gpurun call:
#   bash tools/collect_profiles.sh <tag>        (e.g. r02_a)   -> gpurun_out/profiles_<tag>/
# The command profiled is the driver's own: python3 bench.py --gpus 1 --steps 20 --warmup 5
#   1. that command, plain                                  -> <tag>_bench_driver_cmd.json
#   2. that command under rocprofv3 --kernel-trace --stats  -> <tag>_kernel_stats_driver_cmd.csv + the JSON line of that same run
#   3. that command under --pmc FETCH_SIZE / --pmc WRITE_SIZE (one pass each, no trace domain besides kernel-trace)
#      -> <tag>_pmc_*_c3.csv and traffic_c3_f64_<kernel>.json (tools/traffic_from_pmc.py)
#   4. the other workloads (fp32, C2, C5, C4 planner, front-sweep kernel on C3) as plain bench lines
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5"
python3 $R/bench.py $DRV > $O/${tag}_bench_driver_cmd.json 2> $O/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $DRV > $O/${tag}_bench_driver_cmd_same_run_as_kernel_stats.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_driver_cmd.csv \;
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o pmc -- python3 $R/bench.py $DRV --no-cpu-baseline > /dev/null 2> $O/pmc_$ctr.err
  find $O/pmc_$ctr -name "*counter_collection.csv" -exec cp {} $O/${tag}_pmc_$(echo $ctr | tr A-Z a-z)_c3.csv \;
done
python3 $R/tools/traffic_from_pmc.py $O/${tag}_pmc_fetch_size_c3.csv $O/${tag}_pmc_write_size_c3.csv c3 f64 $O > $O/traffic.log 2>&1
python3 $R/bench.py $DRV --kernel 1 --no-cpu-baseline > $O/${tag}_bench_c3_front_sweep.json 2>/dev/null
python3 $R/bench.py --steps 300 --no-cpu-baseline > $O/${tag}_bench_c3_300steps.json 2>/dev/null
python3 $R/bench.py --dtype f32 --steps 100 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --workload c2 --steps 300 --no-cpu-baseline > $O/${tag}_bench_c2.json 2>/dev/null
python3 $R/bench.py --workload c5 --steps 10 --no-cpu-baseline > $O/${tag}_bench_c5.json 2>/dev/null
python3 $R/bench.py --workload c4 --steps 5 --warmup 1 > $O/${tag}_bench_c4_planner.json 2>/dev/null
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
