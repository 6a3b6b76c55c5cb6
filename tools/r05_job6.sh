#!/bin/bash
# round 5, GPU call 6: priority by what is left of a strip's march (the strips that end last issue first)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job6; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/ab_slowfast.py 24 256 - exp/libvhp_PM512.so exp/libvhp_PM256.so -@pool_contexts=2 exp/libvhp_PM512.so@pool_contexts=4 - > $O/ab_pm.txt 2>&1
timeout 300 python3 tools/unit_timeline.py exp/libvhp_TLPM.so 256 8 > $O/unit_timeline_pm.txt 2>&1
timeout 300 python3 tools/ab_libs.py 4096 128 -@kernel=3 exp/libvhp_PM512.so@kernel=3 > $O/ab_c5.txt 2>&1
timeout 300 python3 tools/ab_libs.py 2048 128 -@kernel=3 exp/libvhp_PM512.so@kernel=3 >> $O/ab_c5.txt 2>&1
tail -8 $O/ab_pm.txt; grep median $O/ab_c5.txt; head -20 $O/unit_timeline_pm.txt
