#!/bin/bash
# round 5, GPU call 9: where the pool sweep (16-step strips) overtakes the front sweep: sides 320-768, 48-1024 sources
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job9; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python3 tools/kernel_ab.py 1,3 48,96,192,384,1024 320x320 384x384 448x448 512x512 576x576 640x640 768x768 1000x1000 > $O/thresholds.txt 2>&1
timeout 600 python3 tools/kernel_ab.py 1,3 48,96,192,384 320x320 448x448 576x576 f32 > $O/thresholds_f32.txt 2>&1
cat $O/thresholds.txt $O/thresholds_f32.txt
