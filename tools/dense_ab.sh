export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
for n in 8 7 6 5; do
  timeout 200 python tools/fwd_ab.py upper $n 45500 262144 sympa_amd/csrc/libsympa_hip.so --flags 4 2>&1 | grep -v amdgpu.ids | sed 's/libsympa_hip.so/old one-launch (flag 4)  /'
  timeout 200 python tools/fwd_ab.py upper $n 45500 262144 sympa_amd/csrc/libsympa_hip.so 2>&1 | grep -v amdgpu.ids | sed 's/libsympa_hip.so/persistent dense        /'
done | tee $O/dense_persistent_ab.txt
timeout 200 python tools/fwd_ab.py upper 8 5041 65536 sympa_amd/csrc/libsympa_hip.so --flags 4 2>&1 | grep -v amdgpu.ids | tee -a $O/dense_persistent_ab.txt
timeout 200 python tools/fwd_ab.py upper 8 5041 65536 sympa_amd/csrc/libsympa_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $O/dense_persistent_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_packed_forward.py -x -q 2>&1 | tail -4 | tee $O/pytest.log
