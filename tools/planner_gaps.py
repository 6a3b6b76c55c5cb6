"""Gaps between consecutive kernels of one plain planner solve in a rocprofv3 kernel trace (the solve that starts at the Nth
vhp_planner_init): sorted, largest first -- the host's polls show up as the large ones.  usage: planner_gaps.py <kernel_trace.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seen, cur = 0, []
for r in rows:
    if "vhp_planner_init" in r["Kernel_Name"]:
        seen += 1
        if seen > n + 1: break
        cur = []
    if seen == n + 1: cur.append(r)
gaps = [((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000, a["Kernel_Name"][:40], b["Kernel_Name"][:40]) for a, b in zip(cur, cur[1:])]
print("%d kernels, span %.0f us, sum of gaps %.0f us" % (len(cur), (int(cur[-1]["End_Timestamp"]) - int(cur[0]["Start_Timestamp"])) / 1000, sum(g[0] for g in gaps)))
g = sorted(gaps, reverse=True)
print("largest:", ["%.1f" % x[0] for x in g[:16]])
print("median %.2f us; gaps above 5 us: %d, their sum %.0f us" % (sorted(x[0] for x in gaps)[len(gaps) // 2], sum(1 for x in gaps if x[0] > 5), sum(x[0] for x in gaps if x[0] > 5)))
