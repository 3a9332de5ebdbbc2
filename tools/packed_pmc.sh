#!/bin/bash
# SQ counters of the packed forward kernels: bash tools/packed_pmc.sh <tag> <n> <nodes> <pairs> <lib.so ...>   (one counter group per run; no trace domains)
TAG=${1:-a}; N=${2:-8}; NODES=${3:-45500}; PAIRS=${4:-262144}; shift 4
OUT=gpurun_out/packed_pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="tools/packed_ab.py upper $N $NODES $PAIRS $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- python3 $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -- python3 $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_IFETCH --output-format csv -d $OUT/sq3 -- python3 $CMD > $OUT/sq3.log 2>&1
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2", "sq3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "packed_forward" not in k: continue
            k = k.split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        for c, x in sorted(v.items()):
            print(f"{d} {k:42s} {c:24s} per launch {x / cnt[(k, c)]:.5g}")
    if not acc:
        print(d, "no rows;", open(f"{out}/{d}.log").read()[-400:])
PY
