#!/bin/bash
# instruction-side counters of the latency sweep at C2 (one source, 1000^2 empty grid): is the kernel's 250 KB of code a cost?
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_c2; mkdir -p $O
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o pmc -- python3 $R/bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/g$i.csv; fi
  rm -rf $O/g$i
done
python3 - <<PY
import csv, glob, collections, statistics as st
for f in sorted(glob.glob("$O/g*.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "vhp_lat_sweep" in r["Kernel_Name"]]
    by=collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in by.items(): print("%-36s median over %d launches: %.0f" % (k, len(v), st.median(v)))
PY
