#!/bin/bash
# round 5, GPU call 2: the 16-step x-major strips (XStrip16): parity, A/B against the 8-step strips (-DVHP_POOL_X8), priorities by phase,
# a longer idle back-off, the floor without stores, the timeline, C5 and other sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job2; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pool.py tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python3 tools/ab_slowfast.py 24 256 - exp/libvhp_X8.so exp/libvhp_PRIO1.so exp/libvhp_SLEEP32.so exp/libvhp_PRIO1S.so exp/libvhp_NOSTORE.so - > $O/ab_slowfast.txt 2>&1
timeout 300 python3 tools/launch_timeline.py exp/libvhp_TL.so r05_b 256 12 > $O/launch_timeline.txt 2>&1
for shape in "4096 128" "2048 128" "1000 96" "1024 256" "640 256"; do set -- $shape
  timeout 300 python3 tools/ab_libs.py $1 $2 -@kernel=3 exp/libvhp_X8.so@kernel=3 exp/libvhp_PRIO1S.so@kernel=3 >> $O/ab_sizes.txt 2>&1
done
AB_DTYPE=f32 timeout 300 python3 tools/ab_libs.py 1000 256 -@kernel=3 exp/libvhp_X8.so@kernel=3 >> $O/ab_sizes.txt 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -3 $O/pytest_gpu.log; tail -9 $O/ab_slowfast.txt; grep median $O/ab_sizes.txt; cut -c1-400 $O/bench_driver_cmd.json
