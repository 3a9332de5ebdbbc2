#!/bin/bash
# PMC passes for the bench workload (one counter group per run, as MI355X_MICROARCH.md prescribes; never combined
# with a trace domain).  usage (on the GPU box): bash tools/pmc_collect.sh <tag> [workload] [kernel substring] [pairs per step] [extra bench args]
# default: the headline command's own kernel -- the fused launch of 32 steps (siegel_dist_multi_kernel)
TAG=${1:-r02}
WORKLOAD=${2:-upper-riem-n4-b65536}
KERNEL=${3:-siegel_dist_multi_kernel}
PPS=${4:-65536}
EXTRA=${5:-}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --workload $WORKLOAD --steps 128 --warmup 32 --no-cpu-baseline --no-live-traffic $EXTRA"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $ARGS > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
python3 tools/pmc_summary.py $OUT "$KERNEL" $PPS | tee $OUT/summary.json
