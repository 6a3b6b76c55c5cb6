#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job7; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python3 tools/launch_timeline.py exp/libvhp_TL.so r05_e 256 12 > $O/launch_timeline.txt 2>&1
grep -A3 "per workgroup" $O/launch_timeline.txt
