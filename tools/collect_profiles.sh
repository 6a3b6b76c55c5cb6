#!/bin/bash
# Collects the artifacts that profiles/ holds for a round, in one gpurun call:
#   bash tools/collect_profiles.sh <tag>        (e.g. r05_a)   -> gpurun_out/profiles_<tag>/
# For the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5 = C3) and for --workload c5:
#   1. the command, plain                                  -> <tag>_bench_<w>.json            (C3: <tag>_bench_driver_cmd.json)
#   2. the command under rocprofv3 --kernel-trace --stats  -> <tag>_kernel_stats_<w>.csv, <tag>_kernel_trace_<w>.csv + the JSON line of that run
#      (the statistics average over every launch of the process -- both timed regions and the allocations probed behind them --; the
#      main region is launches warmup+1 .. warmup+steps of the sweep kernel in the trace: tools/trace_region.py)
#   3. the command under --pmc FETCH_SIZE / --pmc WRITE_SIZE (one pass each, no trace domain besides kernel-trace; the
#      program directly after `--`)                        -> <tag>_pmc_*_<w>.csv and traffic_<w>_f64_<kernel>.json
#   4. the other workloads (fp32, 300 steps, front sweep forced, C2, C4 planner, c1-batch, c-250) as plain bench lines; C2 and C4 also under
#      rocprofv3 --kernel-trace --stats; kernel choice tables; the launch and unit timelines; the slow / fast buffer A/B; SQ counters
# The diagnostic libraries it uses are built HERE first (exp/ is not in the history; the GPU box gets them with the snapshot):
#   for b in "NOSTORE -DVHP_DIAG_NOSTORE" "TL -DVHP_DIAG_TIMELINE" "X8 -DVHP_POOL_X8"; do set -- $b; n=$1; shift; bash tools/build_exp.sh $n "$*"; done
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
profile_workload() {   # $1 = name in file names, $2.. = bench.py arguments
  local w=$1; shift
  python3 $R/bench.py "$@" > $O/${tag}_bench_${w}.json 2> $O/bench_${w}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o kt -- python3 $R/bench.py "$@" > $O/${tag}_bench_${w}_same_run_as_kernel_stats.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_${w}.csv \;
  find $O/kt_$w -name "*kernel_trace.csv" -exec cp {} $O/${tag}_kernel_trace_${w}.csv \;
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${w}_$ctr -o pmc -- python3 $R/bench.py "$@" --no-cpu-baseline > /dev/null 2> $O/pmc_${w}_$ctr.err
    find $O/pmc_${w}_$ctr -name "*counter_collection.csv" -exec cp {} $O/${tag}_pmc_$(echo $ctr | tr A-Z a-z)_${w}.csv \;
  done
  rm -rf $O/kt_$w $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
}
profile_workload driver_cmd --gpus 1 --steps 20 --warmup 5
python3 $R/tools/traffic_from_pmc.py $O/${tag}_pmc_fetch_size_driver_cmd.csv $O/${tag}_pmc_write_size_driver_cmd.csv c3 f64 $O > $O/traffic_c3.log 2>&1
python3 $R/tools/trace_region.py $O/${tag}_kernel_trace_driver_cmd.csv vhp_pool_sweep 5 20 > $O/${tag}_kernel_trace_driver_cmd_timed_region.txt 2>&1
python3 $R/tools/trace_region.py $O/${tag}_kernel_trace_driver_cmd.csv vhp_pool_order 5 20 >> $O/${tag}_kernel_trace_driver_cmd_timed_region.txt 2>&1
profile_workload c5 --gpus 1 --workload c5 --steps 10 --warmup 3
python3 $R/tools/traffic_from_pmc.py $O/${tag}_pmc_fetch_size_c5.csv $O/${tag}_pmc_write_size_c5.csv c5 f64 $O > $O/traffic_c5.log 2>&1
DRV="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
python3 $R/bench.py $DRV --kernel 1 > $O/${tag}_bench_c3_kernel1.json 2>/dev/null
python3 $R/bench.py $DRV --output-buffer placed > $O/${tag}_bench_driver_cmd_output_buffer_placed.json 2>/dev/null
python3 $R/bench.py --steps 300 --no-cpu-baseline > $O/${tag}_bench_c3_300steps.json 2>/dev/null
python3 $R/bench.py --dtype f32 --steps 100 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --workload c2 --steps 300 > $O/${tag}_bench_c2.json 2>/dev/null
python3 $R/bench.py --workload c2 --steps 300 --no-cpu-baseline --kernel 1 > $O/${tag}_bench_c2_kernel1.json 2>/dev/null
python3 $R/bench.py --workload c4 --steps 20 --warmup 2 > $O/${tag}_bench_c4_planner.json 2>/dev/null
python3 $R/bench.py --workload c1-batch --steps 50 --warmup 5 > $O/${tag}_bench_c1_batch.json 2>/dev/null
python3 $R/bench.py --workload c-250 --steps 50 --warmup 5 > $O/${tag}_bench_c_250.json 2>/dev/null
# the latency cases under rocprofv3 --kernel-trace --stats (C2: vhp_lat_sweep; C4: vhp_lat_sweep + vhp_planner_epilogue)
for w in c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o kt -- python3 $R/bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline > $O/${tag}_bench_${w}_same_run_as_kernel_stats.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_${w}.csv \;
  rm -rf $O/kt_$w
done
python3 $R/tools/lat_vs_front.py 256 512 690 1000 1536 2048 > $O/${tag}_lat_vs_front.txt 2>/dev/null
python3 $R/tools/lat_vs_front.py 101 255 689 971 1001 2049 > $O/${tag}_lat_vs_front_odd_widths.txt 2>/dev/null
python3 $R/tools/c1_planner_ab.py > $O/${tag}_planner_odd_widths.txt 2>/dev/null
python3 $R/tools/kernel_ab.py 1,3 32,48,96,192,256 1002x1000 1001x971 690x402 500x500 398x398 250x250 > $O/${tag}_front_vs_pool_other_widths.txt 2>/dev/null
python3 $R/tools/kernel_ab.py 1,3 48,96,192,384,1024 320x320 384x384 448x448 512x512 576x576 640x640 768x768 1000x1000 > $O/${tag}_front_vs_pool_multiples_of_8.txt 2>/dev/null
# the timeline of the C3 launch on the slowest and the fastest of 24 buffers (bytes swept and strips running per 10 us; per workgroup:
# bytes and end time), when every unit was installed and finished, and the kernel against its 8-step build and its no-store build
python3 $R/tools/launch_timeline.py exp/libvhp_TL.so ${tag} 256 24 > $O/${tag}_launch_timeline.txt 2>/dev/null && cp $R/gpurun_out/timeline_${tag}.csv $O/${tag}_launch_timeline.csv
python3 $R/tools/unit_timeline.py exp/libvhp_TL.so 256 24 > $O/${tag}_unit_timeline.txt 2>/dev/null
python3 $R/tools/ab_slowfast.py 40 256 - exp/libvhp_X8.so exp/libvhp_NOSTORE.so - > $O/${tag}_slow_fast_ab.txt 2>/dev/null
for shape in "4096 128" "2048 128" "1000 96" "1024 256" "640 256"; do set -- $shape; python3 $R/tools/ab_libs.py $1 $2 -@kernel=3 exp/libvhp_X8.so@kernel=3 2>/dev/null | grep "^side"; done > $O/${tag}_ab_16_step_vs_8_step_strips_other_sizes.txt
AB_DTYPE=f32 python3 $R/tools/ab_libs.py 1000 256 -@kernel=3 exp/libvhp_X8.so@kernel=3 2>/dev/null | grep "^side" >> $O/${tag}_ab_16_step_vs_8_step_strips_other_sizes.txt
bash $R/tools/pmc_pool.sh > $O/${tag}_pmc_pool_sweep_instruction_side.txt 2>/dev/null
ls -la $O
