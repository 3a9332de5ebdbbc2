#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_collect.sh) for the siegel_dist kernel: per-launch
averages of each counter and the HBM traffic per launch with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads half the
bytes of a wide coalesced read stream: the corrected figure doubles it; both are reported)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "siegel_dist_kernel"
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
avg = {k: v[0] / max(1, v[1]) for k, v in acc.items()}
out = {"counters_avg_per_launch": avg, "launches_sampled": {k: v[1] for k, v in acc.items()}}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    raw = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    corrected = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    out["hbm_bytes_per_launch_raw"] = raw
    out["hbm_bytes_per_launch"] = corrected
if "TCC_HIT_sum" in avg:
    out["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
