#!/bin/bash
# round 5, GPU call 8: small grids -- front sweep against the pool sweep with more contexts per workgroup
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job8; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python3 tools/kernel_ab.py 1,3,3@pool_contexts=6,3@pool_contexts=8@pool_heads=4,3@pool_contexts=12@pool_heads=6 256,1024,4096 104x104 248x248 256x256 384x384 512x512 > $O/small.txt 2>&1
cat $O/small.txt
