#!/bin/bash
# round 5, GPU call 1: parity after the static first round, A/B of the start (static round on/off), priorities by phase, the floor
# without stores, the launch timeline, the SQ counters of the pool sweep at C3, one bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job1; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python3 tools/ab_slowfast.py 24 256 - -@pool_static_round=0 exp/libvhp_PRIO1.so exp/libvhp_PRIO2.so exp/libvhp_NOSTORE.so - > $O/ab_slowfast.txt 2>&1
timeout 300 python3 tools/launch_timeline.py exp/libvhp_TL.so r05_a 256 12 > $O/launch_timeline.txt 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 900 bash tools/pmc_pool.sh > $O/pmc_pool.txt 2>&1
tail -3 $O/pytest_gpu.log; cat $O/ab_slowfast.txt | tail -8; cat $O/bench_driver_cmd.json | cut -c1-600; cat $O/pmc_pool.txt | tail -30
