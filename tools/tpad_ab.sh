mkdir -p gpurun_out/r03u
O=gpurun_out/r03u/tpad_ab.txt
: > $O
for lib in build_ab/pre_tpad.so product build_ab/pre_tpad.so product; do
  echo "== $lib" >> $O
  if [ $lib = product ]; then unset SYMPA_HIP_LIB; else export SYMPA_HIP_LIB=$lib; fi
  python tools/bwd_time_dims.py 2>&1 | grep -v amdgpu.ids >> $O
  DIMS=9,12,16 python tools/dims_time.py 262144 upper 2>&1 | grep -v amdgpu.ids >> $O
  DIMS=12,16 python tools/dims_time.py 262144 bounded 2>&1 | grep -v amdgpu.ids >> $O
  python tools/spd_time.py 16 1048576 100000 --train 2>&1 | grep -v "amdgpu.ids\|generic" >> $O
done
unset SYMPA_HIP_LIB
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 >> $O
cat $O
