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
[ -f profiles/pmc_latest.json ] && python3 tools/pmc_summary.py --help > /dev/null 2>&1 || true
echo copied
