#!/bin/bash
# GRBM_GUI_ACTIVE (core clock cycles the GPU was busy) per sweep launch, over a long run: cycles vs wall time.
out=$1; lib=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
VHP_LIB=${lib:+$R/$lib} rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/$out -o pmc -- python3 $R/bench.py --steps 300 --warmup 3 --no-cpu-baseline "$@" > $R/$out.log 2>&1
python3 - "$R/$out" <<'PY'
import glob, sqlite3, sys
for f in sorted(glob.glob(sys.argv[1] + '/*_results.db')):
    db = sqlite3.connect(f)
    rows = db.execute("select value from counters_collection where kernel_name like '%vhp_sweep_fronts%' and counter_name='GRBM_GUI_ACTIVE'").fetchall()
    v = [r[0] for r in rows]
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = None
    for t in tabs:
        if t == 'kernels' or t.endswith('kernels'):
            try:
                kd = db.execute("select avg(end - start), count(*) from %s where name like '%%vhp_sweep_fronts%%'" % t).fetchone()
                break
            except Exception as e:
                pass
    n = len(v)
    print('launches %d  GUI_ACTIVE/8 per launch: first10 %.0f  last100 %.0f' % (n, sum(v[:10]) / 10 / 8, sum(v[-100:]) / 100 / 8), ' kernel ns avg', kd)
PY
grep -o '"kernel_ms": [0-9.]*' $R/$out.log
