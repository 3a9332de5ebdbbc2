#!/bin/bash
# per-kernel times of the split backward at configs[3]: bash tools/split_prof.sh <out dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-split_prof}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o split -- python3 $GRAFT_REPO_ROOT/tools/bwd_split_ab.py --dims 8 --models upper > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
