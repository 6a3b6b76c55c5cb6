#!/bin/bash
# round 5, GPU call 3: launch-order policy with the 16-step strips and priorities by phase: heads, the small-end share, contexts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job3; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/ab_slowfast.py 24 256 - -@pool_heads=3 -@pool_tail_pct=5 -@pool_tail_pct=2 -@pool_heads=1,pool_tail_pct=5 -@pool_heads=3,pool_contexts=4 -@pool_contexts=4,pool_tail_pct=5 -@pool_claim_ahead=64 -@pool_claim_ahead=24 -@pool_contexts=2 - > $O/ab_policy.txt 2>&1
timeout 300 python3 tools/launch_timeline.py exp/libvhp_TL.so r05_c 256 12 > $O/launch_timeline.txt 2>&1
tail -13 $O/ab_policy.txt
