#!/bin/bash
# SQ / TCP counters of the sweep kernels, one rocprofv3 --pmc pass per counter group (kernel-trace only beside it).
# usage: pmc_stream.sh <out dir under gpurun_out> <lib or -> <kernel option>
out=$1; lib=$2; k=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$out/g$i -o pmc -- python3 $R/tools/one_launch.py $lib $k > $R/gpurun_out/$out/g$i.log 2>&1
done
python3 - "$R/gpurun_out/$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + '/g*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if 'sweep' in row['Kernel_Name'] and 'order' not in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in acc.items():
    print("%-34s mean %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
