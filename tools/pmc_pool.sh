#!/bin/bash
# instruction-side counters of the pool sweep at C3 (256 sources, 1000^2): how busy are the SIMDs' issue ports -- is the launch bound by
# the instructions it issues, by what it waits for, or by the memory?   usage (GPU box): bash tools/pmc_pool.sh [bench args...]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_pool; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o pmc -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --output-buffer first "$@" > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/g$i.csv; fi
  rm -rf $O/g$i
done
python3 - <<PY
import csv, glob, collections, statistics as st
for f in sorted(glob.glob("$O/g*.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "vhp_pool_sweep" in r["Kernel_Name"]]
    by=collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in by.items(): print("%-36s median over %d launches: %.0f" % (k, len(v), st.median(v)))
PY
