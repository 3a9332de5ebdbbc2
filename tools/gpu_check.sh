#!/bin/bash
# One gpurun call: GPU tests, smoke, the numerical self-check of every lanes-per-pair kernel instantiation, bench at the
# driver's K and at the default K, rocprofv3 kernel trace + PMC passes, secondary workloads, training path.
# usage: tools/gpu_check.sh <tag> [notests]
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
FAIL=0
if [ "$2" != "notests" ]; then
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT/pytest.log
grep -q " passed" $OUT/pytest.log && ! grep -q "failed" $OUT/pytest.log || FAIL=1
echo "== smoke" | tee $OUT/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/smoke.log
fi
echo "== self-check gate: every (family, model, n) of the lanes-per-pair kernels against the one-lane kernels" | tee $OUT/selfcheck.txt
timeout 600 python -c "
import torch, sys
from sympa_amd import selfcheck
f = selfcheck.check_all(torch.device('cuda:0'))
print(len(selfcheck.CHECKED), 'instantiations checked,', len(f), 'routed to the one-lane kernels')
for x in f: print('  FALLBACK', x)
sys.exit(1 if f else 0)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/selfcheck.txt
[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
echo "== bench (driver's command line, then defaults)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_k20.err | tail -1 | tee $OUT/bench_k20.json | cut -c1-300
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench.json | cut -c1-300
SYMPA_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic 2>$OUT/bench_k20_dist.err | tail -1 | tee $OUT/bench_k20_forcedist.json | cut -c1-200
echo "== rocprof kernel trace (same commands)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $OUT/prof_k20.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline --no-live-traffic > $OUT/prof_bench.log 2>&1
for d in prof_k20 prof; do find $OUT/$d -name "*kernel_stats.csv" | head -1 | xargs -r head -4; done | tee $OUT/kernel_stats_head.txt
bash tools/pmc_collect.sh $TAG > $OUT/pmc.log 2>&1
tail -30 $OUT/pmc.log
bash tools/pmc_collect.sh ${TAG}_single upper-riem-n4-b65536 "siegel_dist_kernel<4, 0, false" 65536 "--launch graph --streams 1" > $OUT/pmc_single.log 2>&1
tail -30 $OUT/pmc_single.log
echo "== secondary workloads (bench line, then rocprof kernel trace of each, one launch per step and sequential: clean per-kernel figures)"
for W in margulis-bounded-finf-n4-b65536 cartesian-upper-riem-n8-b262144 custom-spd-n16-b1048576 tree-upper-riem-n4-b8192 grid-upper-riem-n2-b512; do
  timeout 600 python bench.py --workload $W --steps 64 --warmup 8 2>$OUT/bench_$W.err | tail -1 > $OUT/bench_$W.json
  cut -c1-300 $OUT/bench_$W.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -- python3 bench.py --workload $W --no-cpu-baseline --no-live-traffic --steps 64 --warmup 8 --launch graph --streams 1 > $OUT/prof_$W.log 2>&1
  find $OUT/prof_$W -name "*kernel_stats.csv" | head -1 | xargs -r head -2
done
echo "== the graph workloads on their real (i < j, d) triplets (--pairs graph: evaluation order; graph-shuffled: training order)"
for W in grid-upper-riem-n2-b512 tree-upper-riem-n4-b8192 margulis-bounded-finf-n4-b65536; do
  for P in graph graph-shuffled; do
    timeout 600 python bench.py --workload $W --pairs $P --steps 64 --warmup 8 --no-cpu-baseline --no-live-traffic 2>>$OUT/bench_graph.err | tail -1 | tee -a $OUT/bench_graph_pairs.json | cut -c1-200
  done
done
echo "== evaluation epoch of the harness (Model.evaluate on configs[1]'s 596 778 triplets, batch 8192)"
timeout 300 python3 tools/eval_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/eval_epoch.txt
echo "== training-path kernels (times, A/Bs, whole graphed step, soak)"
{
  timeout 300 python3 tools/bwd_time_dims.py 2>&1 | grep "fused"
  timeout 300 python3 tools/det_ab.py upper 2>&1 | grep "n="
  timeout 300 python3 tools/det_ab.py bounded 2>&1 | grep "n="
  timeout 300 python3 tools/spd_time.py 16 1048576 100000 --train 2>&1 | grep "spd n="
  timeout 300 python3 tools/table_time.py upper 45500 8 2>&1 | grep rows=
  timeout 300 python3 tools/train_step_time.py 50 2>&1 | grep "training step"
  OPTIM=radam WORKLOADS=grid,tree,margulis,headline timeout 300 python3 tools/train_step_time.py 50 2>&1 | grep "training step"
  timeout 300 python3 tools/fuzz_coop_bwd.py 120 2>&1 | tail -1
  timeout 200 python3 tools/fuzz_coop.py 60 2>&1 | tail -1
} | tee $OUT/training_path.txt
echo "== rocprof kernel trace of the two-kernel training step (headline shape)"
WORKLOADS=headline timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 tools/train_step_time.py 50 > $OUT/prof_train.log 2>&1
find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | tee $OUT/train_kernel_stats_head.txt
echo "== round 4: per-call host cost of the two bindings, --launch direct, the replayed multi-GPU step at world size 1 over RCCL"
timeout 300 python3 tools/host_call_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_call.txt
timeout 300 python bench.py --launch direct --steps 2000 --warmup 200 --no-cpu-baseline --no-live-traffic 2>$OUT/bench_direct.err | tail -1 | tee $OUT/bench_direct.json | cut -c1-260
timeout 600 python3 tools/dist_step_time.py 24 2>&1 | grep "us/step" | tee $OUT/dist_step.txt
echo "== round 4: spd n = 16 backward, three kernels (eigenvectors one pair per lane) against the QL-with-vectors kernel"
{ timeout 600 python3 tools/spd_bwd3_ab.py 1048576 2>&1 | grep -v amdgpu.ids | tail -2; timeout 300 python3 tools/spd_bwd3_ab.py 65536 2>&1 | grep -v amdgpu.ids | tail -2; } | tee $OUT/spd_bwd3_ab.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_spd_bwd3 -- python3 tools/spd_bwd3_ab.py 1048576 > $OUT/prof_spd_bwd3.log 2>&1
find $OUT/prof_spd_bwd3 -name "*kernel_stats.csv" | head -1 | xargs -r head -7 | cut -c1-220
echo "== round 4 (second session): split Siegel backward of dims 7, 8 (one pair per lane, two kernels), merged source rows, atomic rate"
{
  timeout 600 python3 tools/bwd_split_ab.py 2>&1 | grep "^upper\|^bounded"
  echo "# batches sorted by source row (sympa_amd/data.py::sort_batches_by_source):"
  timeout 300 python3 tools/bwd_split_ab.py --dims 7,8 --models upper --sorted 2>&1 | grep "^upper"
  echo "# per-pair rows form:"
  timeout 300 python3 tools/bwd_split_ab.py --dims 7,8 --models upper --rows 2>&1 | grep "^upper"
  echo "# soak against the one-stage kernels:"
  timeout 300 python3 tools/fuzz_split_bwd.py 120 2>&1 | tail -1
} | tee $OUT/split_ab.txt
( cd tools/microbench && hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip > /dev/null 2>&1 ); timeout 120 ./tools/microbench/atomic_rate 2>&1 | tee $OUT/atomic_rate.txt
WORKLOADS=cartesian timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_cartesian -- python3 tools/train_step_time.py 20 > $OUT/prof_train_cartesian.log 2>&1
find $OUT/prof_train_cartesian -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200 | tee $OUT/train_cartesian_kernel_stats_head.txt
WORKLOADS=custom-spd timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_spd -- python3 tools/train_step_time.py 10 > $OUT/prof_train_spd.log 2>&1
find $OUT/prof_train_spd -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200 | tee $OUT/train_spd_kernel_stats_head.txt
timeout 900 bash tools/split_pmc.sh $TAG > $OUT/split_pmc.log 2>&1; tail -75 $OUT/split_pmc.log | cut -c1-160
echo "== round 5: soak of the packed indexed forward, the LDS-DMA micro-benchmark"
timeout 400 python3 tools/fuzz_packed.py 240 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/packed_soak.txt
grep -q "fuzz ok" $OUT/packed_soak.txt || FAIL=1
( cd tools/microbench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds_dma_rate lds_dma_rate.hip > /dev/null 2>&1 ); timeout 120 ./tools/microbench/lds_dma_rate 2>&1 | tee $OUT/lds_dma_rate.txt | tail -4
echo "== round 6: device-side pack validity, measured clock, secondary rows alone, n = 4 backward at larger batches, rows + segmented sum at n = 8"
timeout 300 python3 tools/pack_refresh_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pack_refresh.txt
timeout 300 python3 tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_probe.txt
BUDGET_S=400 timeout 600 python3 tools/bench_rows.py 2>/dev/null | grep -v "^\[" | grep -v amdgpu.ids | tee $OUT/bench_rows.txt
timeout 300 python3 tools/n4_bwd_batches.py 2>&1 | grep "upper n=4" | tee $OUT/n4_bwd_batches.txt
timeout 300 python3 tools/n8_rows_vs_atomics.py 2>&1 | grep "upper n=" | tee $OUT/n8_rows_vs_atomics.txt
echo "== gpu_check status: FAIL=$FAIL"
exit $FAIL
