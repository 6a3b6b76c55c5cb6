#!/bin/bash
# Per-dispatch L2/fabric counters while tools/allocprobe2.py alternates between two output buffers.
out=$1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_WRITE_DRAM_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-50)
  rocprofv3 --pmc $grp --kernel-trace -d $R/$out/$tag -o pmc -- python3 $R/tools/allocprobe2.py > $R/$out/$tag.log 2>&1
  grep "ptr" $R/$out/$tag.log | head -3
done
python3 - "$R/$out" <<'PY'
import glob, sqlite3, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*_results.db')):
    db = sqlite3.connect(f)
    rows = db.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%vhp_sweep_fronts%' order by dispatch_id").fetchall()
    by = {}
    for name, d, v in rows: by.setdefault(name, []).append(v)
    for name, v in by.items():
        # allocprobe2: 43 launches per run(): A, A, B, A, B, A, B, A
        seg = [v[i * 43:(i + 1) * 43] for i in range(8)]
        print("%-40s A: %.4g  A: %.4g  B: %.4g  A: %.4g  B: %.4g" % (name, *[sum(s) / max(len(s), 1) for s in seg[:5]]))
PY
