#!/bin/bash
# Copy the summaries of a tools/gpu_check.sh run (gpurun_out/<tag>/) into profiles/ under their per-round names.
#   tools/copy_profiles.sh r05
TAG=${1:-r05}; SRC=gpurun_out/$TAG; DST=profiles
set -e
cp $SRC/bench_k20.json $DST/${TAG}_bench_k20.json
cp $SRC/bench.json $DST/${TAG}_bench_default.json
cp $SRC/bench_k20_forcedist.json $DST/${TAG}_bench_k20_forcedist.json
cp $SRC/bench_direct.json $DST/${TAG}_bench_direct.json
cp $SRC/bench_graph_pairs.json $DST/${TAG}_bench_graph_pairs.json
for W in margulis-bounded-finf-n4-b65536 cartesian-upper-riem-n8-b262144 custom-spd-n16-b1048576 tree-upper-riem-n4-b8192 grid-upper-riem-n2-b512; do
  cp $SRC/bench_$W.json $DST/${TAG}_bench_$W.json
  f=$(find $SRC/prof_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_${W}_sequential.csv
done
f=$(find $SRC/prof_k20 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_upper_n4_b65536_k20.csv
f=$(find $SRC/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_upper_n4_b65536_default.csv
f=$(find $SRC/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_train_step.csv
f=$(find $SRC/prof_train_cartesian -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_train_step_cartesian.csv
f=$(find $SRC/prof_train_spd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/${TAG}_kernel_stats_train_step_spd.csv
cp $SRC/training_path.txt $DST/${TAG}_training_path.txt
grep "training step" $SRC/training_path.txt > $DST/${TAG}_train_step_times.txt || true
cp $SRC/split_ab.txt $DST/${TAG}_split_backward_ab.txt
cp $SRC/spd_bwd3_ab.txt $DST/${TAG}_spd_backward.txt
cp $SRC/selfcheck.txt $DST/${TAG}_selfcheck.txt
cp $SRC/eval_epoch.txt $DST/${TAG}_eval_epoch.txt
cp $SRC/host_call.txt $DST/${TAG}_host_call.txt
cp $SRC/dist_step.txt $DST/${TAG}_dist_step.txt
tail -3 $SRC/pytest.log > $DST/${TAG}_gpu_pytest_tail.txt
# round 6
for F in pack_refresh clock_probe bench_rows n4_bwd_batches n8_rows_vs_atomics packed_soak atomic_rate lds_dma_rate; do
  [ -f $SRC/$F.txt ] && cp $SRC/$F.txt $DST/${TAG}_$F.txt
done
# the PMC pass of the headline command (tools/pmc_collect.sh <tag>): per-round summary + the entry bench.py falls back to
if [ -f gpurun_out/pmc_$TAG/summary.json ]; then
  cp gpurun_out/pmc_$TAG/summary.json $DST/${TAG}_pmc_summary.json
  python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
cur = json.load(open("profiles/pmc_latest.json"))
new = json.load(open(f"gpurun_out/pmc_{tag}/summary.json"))
new["round"] = tag
new["source"] = (f"tools/pmc_collect.sh {tag} (rocprofv3 --pmc, one counter group per pass, `bench.py --steps 128 --warmup 32`: fused "
                 "launches of 32 steps, every counter normalised by the work-items of its dispatches)")
cur["upper-riem-n4-b65536"] = new
json.dump(cur, open("profiles/pmc_latest.json", "w"), indent=1)
PY
fi
echo copied
