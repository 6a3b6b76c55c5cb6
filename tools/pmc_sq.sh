#!/bin/bash
# Collect SQ instruction/issue counters for the default bench workload (one counter group per pass).
# Usage (on the GPU box): bash tools/pmc_sq.sh <outdir> [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
DEFAULT_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY;SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_ANY;GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
IFS=";" read -ra GRPS <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GRPS[@]}"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $grp --kernel-trace -d $R/$out/$tag -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/$out/$tag.log 2>&1
done
python3 - "$R/$out" <<'PY'
import glob, sqlite3, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*_results.db')):
    db = sqlite3.connect(f)
    for name, tot, n in db.execute("select counter_name, sum(value), count(*) from counters_collection "
                                   "where kernel_name like '%vhp_sweep_fronts%' group by counter_name"):
        print('%-32s per launch %.5g  (launches %d)' % (name, tot / max(n, 1), n))
PY
