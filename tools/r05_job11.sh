#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job11; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/ab_slowfast.py 24 256 - -@pool_static_round=2 -@pool_static_round=2,pool_heads=3 -@pool_static_round=2,pool_tail_pct=5 - -@pool_static_round=2 > $O/ab_snake.txt 2>&1
tail -9 $O/ab_snake.txt
