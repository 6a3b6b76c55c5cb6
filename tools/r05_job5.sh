#!/bin/bash
# round 5, GPU call 5: (a) on slow memory the first units installed -- the longest -- take the whole launch: does a cap on the wavefronts
# that sweep while units are installed (busy_cap) get them through earlier?  (b) C5: 16-step y-major strips against 8-step ones
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job5; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/ab_slowfast.py 24 256 - -@pool_busy_cap=10 -@pool_busy_cap=9 -@pool_busy_cap=8 -@pool_busy_cap=7 -@pool_busy_cap=6 -@pool_busy_cap=8,pool_heads=3 -@pool_busy_cap=8,pool_contexts=4 -@pool_busy_cap=6,pool_contexts=4 - > $O/ab_busy.txt 2>&1
timeout 600 python3 tools/ab_libs.py 4096 128 -@kernel=3 exp/libvhp_X8.so@kernel=3 exp/libvhp_Y8.so@kernel=3 -@kernel=3,pool_contexts=2 > $O/ab_c5.txt 2>&1
tail -12 $O/ab_busy.txt; grep median $O/ab_c5.txt
