cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_spec
timeout 300 python3 -m pytest tests/test_gpu_speculative.py tests/test_gpu_planner.py -x -q > gpurun_out/r05_spec/pytest5.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r05_spec/pytest5.txt | tail -3
timeout 200 python3 tools/spec_modes.py 5 > gpurun_out/r05_spec/modes.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_spec/prof2 -o m -- python3 tools/spec_modes.py 1 > /dev/null 2>&1
python3 tools/spec_trace_split.py gpurun_out/r05_spec/prof2/m_kernel_trace.csv 2 > gpurun_out/r05_spec/split.txt 2>&1; rm -rf gpurun_out/r05_spec/prof2
cat gpurun_out/r05_spec/modes.txt gpurun_out/r05_spec/split.txt
