#!/bin/bash
# instruction-cache counters of the split backward kernels: bash tools/split_pmc_icache.sh <tag>
TAG=${1:-a}
OUT=gpurun_out/split_icache_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="tools/bwd_split_ab.py --dims 8 --models upper --sorted --split-only"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/ic -- python3 $CMD > $OUT/ic.log 2>&1
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "siegel_bwd_gradient" not in k and "siegel_bwd_spectral" not in k: continue
        k = k.split("(")[0][-45:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    for c, x in sorted(v.items()):
        print(f"{k:46s} {c:28s} per launch {x / cnt[(k, c)]:.5g}")
if not acc: print("no rows;", open(f"{out}/ic.log").read()[-600:])
PY
