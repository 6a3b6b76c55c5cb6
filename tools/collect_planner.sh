cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_d; mkdir -p $O
python3 bench.py --workload c4 --steps 20 --warmup 2 > $O/r05_d_bench_c4_planner.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --workload c4 --steps 20 --warmup 2 --no-cpu-baseline > $O/r05_d_bench_c4_same_run_as_kernel_stats.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/r05_d_kernel_stats_c4.csv \;
rm -rf $O/kt $O/kt.err
python3 tools/spec_modes.py 5 > $O/r05_d_planner_modes_maze6.txt 2>&1
python3 tools/spec_modes.py 3 0.25 250 > $O/r05_d_planner_modes_maze6_threshold_0.25_livelock.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -o m -- python3 tools/spec_modes.py 1 > /dev/null 2>&1
python3 tools/spec_trace_split.py $O/kt2/m_kernel_trace.csv 2 > $O/r05_d_planner_modes_kernel_durations.txt 2>&1
rm -rf $O/kt2
python3 bench.py --workload c2 --steps 300 > $O/r05_d_bench_c2.json 2>/dev/null
python3 tools/c1_planner_ab.py > $O/r05_d_planner_odd_widths.txt 2>/dev/null
ls -la $O
