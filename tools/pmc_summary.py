#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_collect.sh) for one kernel:
    python tools/pmc_summary.py <dir> <kernel substring> [pairs per step]
A bench run launches the kernel with several grid sizes (fused launches of 20 / 32 steps, single steps), so every
counter is normalised by the work-items of the dispatches it was sampled on (Grid_Size = pairs, one pair per lane;
sixteen lanes per pair for the *_coop kernels) and reported per STEP of `pairs per step` pairs.  HBM traffic with the
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads
half the bytes of a 16 B/lane read stream, the corrected figure doubles it; both are reported."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "siegel_dist_kernel"
pairs_per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
lanes_per_pair = 16 if "coop" in kernel else 1
acc = defaultdict(lambda: [0.0, 0, 0.0])      # counter -> [sum of values, dispatches, sum of grid sizes]
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            a[2] += float(row["Grid_Size"])
per_step = {k: v[0] / max(1.0, v[2]) * lanes_per_pair * pairs_per_step for k, v in acc.items()}
out = {"kernel": kernel, "pairs_per_step": pairs_per_step,
       "counters_per_step": per_step,
       "dispatches_sampled": {k: v[1] for k, v in acc.items()},
       "avg_pairs_per_dispatch": {k: v[2] / lanes_per_pair / max(1, v[1]) for k, v in acc.items()}}
# (kept under the old name too: bench.py / older summaries read counters_avg_per_launch of a one-step launch)
out["counters_avg_per_launch"] = per_step
if "FETCH_SIZE" in per_step and "WRITE_SIZE" in per_step:
    out["hbm_bytes_per_step_raw"] = (per_step["FETCH_SIZE"] + per_step["WRITE_SIZE"]) * 1024.0
    out["hbm_bytes_per_step"] = (2.0 * per_step["FETCH_SIZE"] + per_step["WRITE_SIZE"]) * 1024.0
if "TCC_HIT_sum" in per_step:
    out["l2_hit_rate"] = per_step["TCC_HIT_sum"] / max(1.0, per_step["TCC_HIT_sum"] + per_step["TCC_MISS_sum"])
if per_step.get("SQ_WAVES"):
    out["valu_insts_per_wave"] = per_step["SQ_INSTS_VALU"] / per_step["SQ_WAVES"]
    out["wait_any_frac_of_wave_cycles"] = per_step["SQ_WAIT_ANY"] / per_step["SQ_WAVE_CYCLES"]
    out["valu_active_frac_of_wave_cycles"] = per_step["SQ_ACTIVE_INST_VALU"] / per_step["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1))
