"""HBM bytes per launch of the batch-sweep kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
usage: traffic_from_pmc.py <fetch csv> <write csv> <workload> <dtype> <out dir>
Writes traffic_<workload>_<dtype>_<kernel>.json, which bench.py quotes (labelled static) as roofline.traffic.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): WRITE_SIZE is exact for 16-byte-per-lane streaming stores;
FETCH_SIZE counts a wide coalesced read at half its bytes -- the sweep's reads are 8-byte occupancy words and table entries,
not that pattern, so both the raw and the doubled figure are recorded and the doubled one (the upper bound) is used."""
import csv, json, os, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch_csv, write_csv, workload, dtype, out_dir = sys.argv[1:6]
fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
KERNELS = ("vhp_sweep_fronts", "vhp_pool_sweep", "vhp_lat_sweep")   # the batch-sweep kernels (bench.py names them the same way)
names = [k for k in write if any(n in k for n in KERNELS)]
name = max(names, key=lambda k: sum(write[k]))
short = next(n for n in KERNELS if n in name)
w = sum(write[name]) / len(write[name])
fch = sum(fetch[name]) / len(fetch[name])
out = {
    "workload": workload, "dtype": dtype, "kernel": short, "launches_averaged": len(write[name]),
    "FETCH_SIZE_KB_per_launch": fch, "WRITE_SIZE_KB_per_launch": w,
    "hbm_bytes_per_launch": (2.0 * fch + w) * 1024.0,
    "hbm_bytes_per_launch_fetch_uncorrected": (fch + w) * 1024.0,
    "source": [os.path.basename(fetch_csv), os.path.basename(write_csv)],
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the bench command of this workload (tools/collect_profiles.sh); "
            "mean over the launches of the process; FETCH_SIZE doubled (gfx950 correction, an upper bound for this read pattern)",
}
path = os.path.join(out_dir, "traffic_%s_%s_%s.json" % (workload, dtype, short))
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out))
