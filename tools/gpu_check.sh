#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprofv3 kernel trace + PMC passes.   usage: tools/gpu_check.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT/pytest.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
echo "== bench" | tee $OUT/bench.log
timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json
echo "== rocprof kernel trace (same command)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline > $OUT/prof_bench.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -3 | tee $OUT/kernel_stats_head.txt
bash tools/pmc_collect.sh $TAG > $OUT/pmc.log 2>&1
tail -25 $OUT/pmc.log
echo "== secondary workloads (rocprof kernel trace of each)"
for W in margulis-bounded-finf-n4-b65536 cartesian-upper-riem-n8-b262144 custom-spd-n16-b1048576; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 bench.py --workload $W --no-cpu-baseline --steps 64 --warmup 8 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  tail -1 $OUT/bench_$W.json | cut -c1-400
  find $OUT/prof_$W -name "*kernel_stats.csv" | head -1 | xargs -r head -2
done
