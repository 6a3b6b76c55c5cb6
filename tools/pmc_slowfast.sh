#!/bin/bash
# counters of the C3 launch on a slow and on a fast buffer of one process; one rocprofv3 pass per counter group
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_sf; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum" "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_PROBE_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o pmc -- python3 $R/tools/pmc_slowfast.py 16 > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/g$i.csv; fi
  rm -rf $O/g$i
  tail -2 $O/g$i.log
done
