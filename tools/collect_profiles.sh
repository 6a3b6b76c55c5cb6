#!/bin/bash
# Collects the artifacts that profiles/ holds for a round, in one gpurun call:
#   bash tools/collect_profiles.sh <tag>        (e.g. r01_e)   -> gpurun_out/profiles_<tag>/
# bench JSON lines, the rocprofv3 kernel-trace summary of the same default command, and the HBM traffic
# counters (FETCH_SIZE and WRITE_SIZE, one --pmc pass each, no trace domains besides kernel-trace).
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench_c3_default.json 2> $O/bench_default.err
python3 $R/bench.py --steps 300 --no-cpu-baseline > $O/${tag}_bench_c3_300steps.json 2>/dev/null
python3 $R/bench.py --dtype f32 --steps 100 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --workload c2 --steps 300 --no-cpu-baseline > $O/${tag}_bench_c2.json 2>/dev/null
python3 $R/bench.py --workload c5 --steps 10 --no-cpu-baseline > $O/${tag}_bench_c5.json 2>/dev/null
python3 $R/tools/bench_planner.py > $O/${tag}_planner.txt 2>&1
# kernel trace of the default bench command without the placement probes (so that every sweep launch of the
# process belongs to the warm-up or to the timed region; the JSON of this very run is kept beside the stats)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --placements 1 > $O/${tag}_bench_c3_same_run_as_kernel_stats.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_c3_profiled_bench.csv \;
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --placements 1 > /dev/null 2> $O/pmc_$ctr.err
  find $O/pmc_$ctr -name "*counter_collection.csv" -exec cp {} $O/${tag}_pmc_$(echo $ctr | tr A-Z a-z)_c3.csv \;
done
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
