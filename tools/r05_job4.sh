#!/bin/bash
# round 5, GPU call 4: 16-step y-major strips (YStrip16): parity, A/B against the 8-step strips, per-unit timeline, other sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job4; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pool.py tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python3 tools/ab_slowfast.py 24 256 - exp/libvhp_X8.so exp/libvhp_NOSTORE.so - > $O/ab_slowfast.txt 2>&1
timeout 300 python3 tools/unit_timeline.py exp/libvhp_TL.so 256 8 > $O/unit_timeline.txt 2>&1
timeout 300 python3 tools/launch_timeline.py exp/libvhp_TL.so r05_d 256 12 > $O/launch_timeline.txt 2>&1
for shape in "4096 128" "2048 128" "1000 96" "1024 256" "640 256"; do set -- $shape
  timeout 300 python3 tools/ab_libs.py $1 $2 -@kernel=3 exp/libvhp_X8.so@kernel=3 >> $O/ab_sizes.txt 2>&1
done
AB_DTYPE=f32 timeout 300 python3 tools/ab_libs.py 1000 256 -@kernel=3 exp/libvhp_X8.so@kernel=3 >> $O/ab_sizes.txt 2>&1
tail -3 $O/pytest_gpu.log; tail -6 $O/ab_slowfast.txt; grep median $O/ab_sizes.txt; cat $O/unit_timeline.txt
