cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_kt; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --workload c4 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c4.csv \;
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_c4.csv \;
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_kt/kernel_trace_c4.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find a run of alternating lat/epilogue in the middle
import collections
seq=[(r['Kernel_Name'][:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
idx=[i for i,s in enumerate(seq) if 'vhp_lat_sweep' in s[0]]
mid=idx[len(idx)//2]
for i in range(mid, mid+12):
    n,s,e=seq[i]; pn,ps,pe=seq[i-1]
    print("%-40s dur %6.2f us  gap before %6.2f us" % (n,(e-s)/1e3,(s-pe)/1e3))
PY
head -12 $O/kernel_stats_c4.csv | cut -c1-160
rm -rf $O/kt
