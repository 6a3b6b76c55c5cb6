"""The timed region of a bench.py run inside a rocprofv3 kernel trace: mean duration of launches warmup+1 .. warmup+steps of the
sweep kernel (the launches before them are warm-up, the ones after them time other buffers: config.output_placement).
usage: trace_region.py <kernel_trace.csv> <kernel name part> <warmup> <steps>"""
import csv, sys
path, name, warm, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
reg = d[warm: warm + steps]
print("%d launches of %s; timed region (launches %d..%d): mean %.1f us, min %.1f, max %.1f; all launches: mean %.1f us" % (
    len(d), name, warm + 1, warm + steps, sum(reg) / len(reg), min(reg), max(reg), sum(d) / len(d)))
