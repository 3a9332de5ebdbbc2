OUT=gpurun_out/r04e
{
  timeout 300 python3 tools/bwd_time_dims.py 2>&1 | grep "fused"
  timeout 300 python3 tools/det_ab.py upper 2>&1 | grep "n="
  timeout 300 python3 tools/det_ab.py bounded 2>&1 | grep "n="
  timeout 300 python3 tools/spd_time.py 16 1048576 100000 --train 2>&1 | grep "spd n="
  timeout 300 python3 tools/table_time.py upper 45500 8 2>&1 | grep rows=
  cat $OUT/train_step_times.txt
  OPTIM=radam WORKLOADS=grid,tree,margulis,headline timeout 300 python3 tools/train_step_time.py 50 2>&1 | grep "training step"
  timeout 300 python3 tools/fuzz_coop_bwd.py 120 2>&1 | tail -1
  timeout 200 python3 tools/fuzz_coop.py 60 2>&1 | tail -1
} | tee $OUT/training_path.txt
