#!/bin/bash
# The headline lines of a round in one gpurun call: the driver's command and C5, plain and under rocprofv3 --kernel-trace --stats,
# fp32, 300 steps, --output-buffer first.  usage: bash tools/collect_headline.sh   (-> gpurun_out/profiles_<tag>/; tag below)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r04_f; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
tag=r04_f
pw() { w=$1; shift
  python3 $R/bench.py "$@" > $O/${tag}_bench_${w}.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o kt -- python3 $R/bench.py "$@" > $O/${tag}_bench_${w}_same_run_as_kernel_stats.json 2>/dev/null
  find $O/kt_$w -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_${w}.csv \;
  # (every launch of the run: the statistics average over the timed buffer, the first allocation and three more; the timed region is
  # launches warmup+1 .. warmup+steps of the sweep kernel in this file, tools/trace_region.py)
  find $O/kt_$w -name "*kernel_trace.csv" -exec cp {} $O/${tag}_kernel_trace_${w}.csv \;
  rm -rf $O/kt_$w
}
pw driver_cmd --gpus 1 --steps 20 --warmup 5
pw c5 --gpus 1 --workload c5 --steps 10 --warmup 3 --no-cpu-baseline
python3 $R/bench.py --dtype f32 --steps 100 --no-cpu-baseline > $O/${tag}_bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --output-buffer first --steps 20 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_driver_cmd_output_buffer_first.json 2>/dev/null
python3 $R/bench.py --steps 300 --no-cpu-baseline > $O/${tag}_bench_c3_300steps.json 2>/dev/null
ls -la $O
