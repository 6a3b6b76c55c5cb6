#!/bin/bash
# Round 6's profile collection, one gpurun call:  bash tools/collect_r06.sh <tag>   -> gpurun_out/profiles_<tag>/
# (tools/collect_profiles.sh is the full collection of rounds 4-5: C3 / C5 with their PMC passes and the pool sweep's timelines; this
# one adds what round 6 changed -- the latency sweep in bands, the planner's loop on it, small batches, the union kernel -- and repeats
# the driver's command and C5 with kernel statistics so that the headline's files are of this tree.)
# The diagnostic libraries are built first, here:  bash tools/build_exp.sh STRIPS "-DVHP_LAT_STRIPS";  bash tools/build_exp.sh PPW "-DVHP_DIAG_POOLPROF -DVHP_DIAG_WINPROF"
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # $1 = name in file names, $2.. = bench.py arguments: the command plain, then under rocprofv3 --kernel-trace --stats
  local w=$1; shift
  python3 $R/bench.py "$@" > $O/${tag}_bench_${w}.json 2> $O/bench_${w}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o kt -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/${tag}_bench_${w}_same_run_as_kernel_stats.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_${w}.csv \;
  find $O/kt_$w -name "*kernel_trace.csv" -exec cp {} $O/${tag}_kernel_trace_${w}.csv \;
  rm -rf $O/kt_$w
}
pmc() {     # $1 = name, $2.. = bench.py arguments: FETCH_SIZE and WRITE_SIZE, one pass each, no trace domain besides kernel-trace
  local w=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${w}_$ctr -o pmc -- python3 $R/bench.py "$@" --no-cpu-baseline > /dev/null 2> $O/pmc_${w}_$ctr.err
    find $O/pmc_${w}_$ctr -name "*counter_collection.csv" -exec cp {} $O/${tag}_pmc_$(echo $ctr | tr A-Z a-z)_${w}.csv \;
    rm -rf $O/pmc_${w}_$ctr
  done
}
stats driver_cmd --gpus 1 --steps 20 --warmup 5
python3 $R/tools/trace_region.py $O/${tag}_kernel_trace_driver_cmd.csv vhp_pool_sweep 5 20 > $O/${tag}_kernel_trace_driver_cmd_timed_region.txt 2>&1
python3 $R/tools/trace_region.py $O/${tag}_kernel_trace_driver_cmd.csv vhp_pool_order 5 20 >> $O/${tag}_kernel_trace_driver_cmd_timed_region.txt 2>&1
pmc driver_cmd --gpus 1 --steps 20 --warmup 5
python3 $R/tools/traffic_from_pmc.py $O/${tag}_pmc_fetch_size_driver_cmd.csv $O/${tag}_pmc_write_size_driver_cmd.csv c3 f64 $O > $O/traffic_c3.log 2>&1
stats c5 --gpus 1 --workload c5 --steps 10 --warmup 3
stats c2 --workload c2 --steps 300 --warmup 10
stats c5_8_sources --workload c5 --sources 8 --steps 50 --warmup 5   # (4096^2 with 8 sources: the latency sweep with four workgroups per octant)
stats c4 --workload c4 --steps 20 --warmup 2
for n in 8 32 64 96; do
  stats c3_$n --workload c3-$n --steps 100 --warmup 10
done
pmc c3_32 --workload c3-32 --steps 100 --warmup 10
python3 $R/bench.py --workload c1-batch --steps 50 --warmup 5 > $O/${tag}_bench_c1_batch.json 2>/dev/null
python3 $R/bench.py --workload c-250 --steps 50 --warmup 5 > $O/${tag}_bench_c_250.json 2>/dev/null
python3 $R/bench.py --dtype f32 --steps 100 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --output-buffer placed > $O/${tag}_bench_driver_cmd_output_buffer_placed.json 2>/dev/null
rm -f $O/${tag}_kernel_trace_c5.csv $O/${tag}_kernel_trace_c3_*.csv $O/${tag}_kernel_trace_c2.csv
# the planner's loop: sweep / epilogue durations of the plain loop, the modes
python3 - > $O/${tag}_planner_loop_kernel_durations.txt 2>&1 <<PY
import csv, numpy as np
rows = sorted(csv.DictReader(open("$O/${tag}_kernel_trace_c4.csv")), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
pairs = [i for i in range(len(seq) - 1) if "vhp_lat_sweep" in seq[i][0] and "vhp_planner_epilogue" in seq[i + 1][0]]
sw = np.array([(seq[i][2] - seq[i][1]) / 1e3 for i in pairs]); ep = np.array([(seq[i + 1][2] - seq[i + 1][1]) / 1e3 for i in pairs])
g1 = np.array([(seq[i + 1][1] - seq[i][2]) / 1e3 for i in pairs]); g0 = np.array([(seq[i][1] - seq[i - 1][2]) / 1e3 for i in pairs])
live = sw > 3
print("plain planner loop on maze_6 under rocprofv3 --kernel-trace (bench.py --workload c4): %d iterations that swept" % live.sum())
print("  vhp_lat_sweep       mean %.2f us, median %.2f, 10th / 90th percentile %.2f / %.2f" % (sw[live].mean(), np.median(sw[live]), *np.percentile(sw[live], [10, 90])))
print("  vhp_planner_epilogue mean %.2f us" % ep[live].mean())
print("  between sweep and epilogue %.2f us, between epilogue and the next sweep %.2f us (medians; the profiler's own serialisation included)" % (np.median(g1[live]), np.median(g0[live])))
PY
rm -f $O/${tag}_kernel_trace_c4.csv
python3 $R/tools/spec_modes.py 5 > $O/${tag}_planner_modes_maze6.txt 2>&1
# the latency sweep against the front sweep by grid and batch; bands against the strips of rounds 3-5 on the same box
python3 $R/tools/lat_vs_front.py 256 512 690 1000 1536 2048 > $O/${tag}_lat_vs_front.txt 2>/dev/null
python3 $R/tools/lat_vs_front.py 101 255 689 971 1001 2049 > $O/${tag}_lat_vs_front_odd_widths.txt 2>/dev/null
{ python3 $R/tools/ab_libs.py 0 1 - exp/libvhp_STRIPS.so; for s in "256 1" "512 1" "690 1" "1000 8" "1000 32" "2048 4" "4096 2"; do set -- $s; python3 $R/tools/ab_libs.py $1 $2 -@kernel=4 exp/libvhp_STRIPS.so@kernel=4; done; } 2>/dev/null | grep "^side" > $O/${tag}_ab_bands_vs_strips.txt
python3 $R/tools/lat_timeline.py exp/libvhp_PPW.so 1000 > $O/${tag}_lat_timeline_c2.txt 2>/dev/null
python3 $R/tools/lat_timeline.py exp/libvhp_PPW.so 0 345 391 > $O/${tag}_lat_timeline_maze6_start.txt 2>/dev/null
python3 $R/tools/c1_planner_ab.py > $O/${tag}_planner_odd_widths.txt 2>/dev/null
python3 $R/tools/kernel_ab.py 1,3 32,48,96,192,256 1002x1000 1001x971 690x402 > $O/${tag}_front_vs_pool_other_widths.txt 2>/dev/null
ls -la $O
