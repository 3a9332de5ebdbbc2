#!/usr/bin/env python3
"""All-pairs distance matrix (runner.py:142-154): packed kernel vs the pairwise kernel in index-free mode.
   python tools/allpairs_time.py [N] [n] [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5041
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = sys.argv[3] if len(sys.argv) > 3 else "upper"
dev = torch.device("cuda:0")
table = data.trained_like_table(N, n, model=model, seed=1).to(dev)
out = torch.empty(N, N, dtype=torch.float64, device=dev)
lib_ws = torch.empty(max(1, ops._lib.load().sympa_all_pairs_workspace_bytes(N, n, ops.MODEL_IDS[model]) // 8),
                     dtype=torch.float64, device=dev)
for packed, fl in ((False, 0), (True, ops.FLAG_NO_SYMMETRY), (True, 0)):
    for _ in range(2):
        ops.all_pairs_dist(table, model, "riem", out=out, packed=packed, workspace=lib_ws if packed else None, flags=fl)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.all_pairs_dist(table, model, "riem", out=out, packed=packed, workspace=lib_ws if packed else None, flags=fl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{model} n={n} N={N} packed={packed} no_symmetry={bool(fl)}: {dt * 1e3:8.3f} ms  {N * N / dt / 1e9:7.2f} G matrix entries/s")
ops.check_status(dev)
