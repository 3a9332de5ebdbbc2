#!/bin/bash
# The short form of tools/gpu_check.sh: GPU tests, smoke, the driver's bench line, the configs[3] / configs[4] bench lines and their
# sequential rocprofv3 kernel stats.   usage: tools/gpu_check_short.sh <tag>
TAG=${1:-r05s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT/pytest.log
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_k20.err | tail -1 | tee $OUT/bench_k20.json | cut -c1-300
for W in cartesian-upper-riem-n8-b262144 custom-spd-n16-b1048576; do
  timeout 600 python bench.py --workload $W --steps 64 --warmup 8 2>$OUT/bench_$W.err | tail -1 > $OUT/bench_$W.json
  cut -c1-300 $OUT/bench_$W.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 bench.py --workload $W --no-cpu-baseline --no-live-traffic --steps 64 --warmup 8 --launch graph --streams 1 > $OUT/prof_$W.log 2>&1
  find $OUT/prof_$W -name "*kernel_stats.csv" | head -1 | xargs -r head -3 | cut -c1-200
done
WORKLOADS=cartesian timeout 300 python3 tools/train_step_time.py 20 2>&1 | grep "training step" | tee $OUT/train_cartesian.txt
