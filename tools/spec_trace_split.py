"""Splits a rocprofv3 kernel trace of tools/spec_modes.py per solve (a solve starts with vhp_planner_init) and prints, per solve, the
count / mean / median duration (us) of the sweep and epilogue kernels and the mean gap between consecutive kernels.
usage: spec_trace_split.py <kernel_trace.csv> [every Nth solve]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 1
solves, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "vhp_planner_init" in n:
        cur = []
        solves.append(cur)
    if cur is not None and any(k in n for k in ("lat_sweep", "epilogue", "spec_sweep", "planner_sweep")):
        cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for i, s in enumerate(solves):
    if i % step or not s: continue
    acc = collections.defaultdict(list)
    gaps = []
    for j, (n, t0, t1) in enumerate(s):
        key = "sweep" if "sweep" in n else ("spec_epilogue" if "spec_epilogue" in n else "epilogue")
        acc[key].append((t1 - t0) / 1000)
        if j: gaps.append((t0 - s[j - 1][2]) / 1000)
    print("solve %3d: " % i + "  ".join("%s n=%d mean %.1f median %.1f" % (k, len(v), sum(v) / len(v), sorted(v)[len(v) // 2]) for k, v in sorted(acc.items()))
          + "  gap mean %.1f" % (sum(gaps) / max(len(gaps), 1)) + "  span %.0f us" % ((s[-1][2] - s[0][1]) / 1000))
