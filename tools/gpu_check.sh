#!/bin/bash
# One gpurun call: GPU tests, smoke, bench at the driver's K and at the default K, rocprofv3 kernel trace + PMC passes.
# usage: tools/gpu_check.sh <tag> [notests]
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT/pytest.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
fi
echo "== bench (driver's command line, then defaults)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_k20.err | tail -1 | tee $OUT/bench_k20.json | cut -c1-300
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench.json | cut -c1-300
echo "== rocprof kernel trace (same commands)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_k20.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline > $OUT/prof_bench.log 2>&1
for d in prof_k20 prof; do find $OUT/$d -name "*kernel_stats.csv" | head -1 | xargs -r head -4; done | tee $OUT/kernel_stats_head.txt
bash tools/pmc_collect.sh $TAG > $OUT/pmc.log 2>&1
tail -30 $OUT/pmc.log
bash tools/pmc_collect.sh ${TAG}_single upper-riem-n4-b65536 "siegel_dist_kernel<4, 0, false" 65536 "--launch graph --streams 1" > $OUT/pmc_single.log 2>&1
tail -30 $OUT/pmc_single.log
echo "== secondary workloads (bench line, then rocprof kernel trace of each, one launch per step and sequential: clean per-kernel figures)"
for W in margulis-bounded-finf-n4-b65536 cartesian-upper-riem-n8-b262144 custom-spd-n16-b1048576 tree-upper-riem-n4-b8192 grid-upper-riem-n2-b512; do
  timeout 600 python bench.py --workload $W --steps 64 --warmup 8 2>$OUT/bench_$W.err | tail -1 > $OUT/bench_$W.json
  cut -c1-300 $OUT/bench_$W.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 bench.py --workload $W --no-cpu-baseline --steps 64 --warmup 8 --launch graph --streams 1 > $OUT/prof_$W.log 2>&1
  find $OUT/prof_$W -name "*kernel_stats.csv" | head -1 | xargs -r head -2
done
echo "== training-path kernels (times, A/B against the one-lane kernels, whole graphed step, soak)"
{
  timeout 300 python3 tools/bwd_time_dims.py 2>&1 | grep "fused"
  timeout 300 python3 tools/bwd_coop_ab_small.py 262144 upper 2>&1 | grep "n="
  timeout 300 python3 tools/bwd_coop_ab_small.py 262144 bounded 2>&1 | grep "n="
  timeout 300 python3 tools/bwd_coop_ab.py upper 65536 9 10 12 16 2>&1 | grep backward
  timeout 300 python3 tools/bwd_coop_ab.py bounded 65536 9 10 12 16 2>&1 | grep backward
  timeout 300 python3 tools/spd_bwd_ab.py 65536 4 8 12 16 2>&1 | grep "spd backward"
  timeout 300 python3 tools/spd_time.py 16 1048576 100000 --train 2>&1 | grep "spd n="
  timeout 300 python3 tools/table_time.py upper 45500 8 2>&1 | grep rows=
  timeout 300 python3 tools/table_time.py upper 5041 10 16 2>&1 | grep rows=
  timeout 300 python3 tools/train_step_time.py 20 2>&1 | grep "training step"
  timeout 300 python3 tools/fuzz_coop_bwd.py 120 2>&1 | tail -1
  timeout 200 python3 tools/fuzz_coop.py 60 2>&1 | tail -1
} | tee $OUT/training_path.txt
