#!/usr/bin/env python3
"""Forward throughput by dims (riem): python tools/dims_time.py [batch] [upper|bounded]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = sys.argv[2] if len(sys.argv) > 2 else "upper"
dev = torch.device("cuda:0")
for n in [int(x) for x in os.environ.get("DIMS", "2,3,4,5,6,7,8,9,10,11,12,13,14,15,16").split(",")]:
    nodes = 5000
    table = data.trained_like_table(nodes, n, seed=1, model=model).to(dev)
    pairs = data.sample_pairs(nodes, b, 0, 1).to(dev)
    out = torch.empty(b, dtype=torch.float64, device=dev)
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    for _ in range(3):
        ops.model_forward(table, pairs, model, "riem", None, scale, 1.0, out=out)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.model_forward(table, pairs, model, "riem", None, scale, 1.0, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    bpp = 32 * n * n + 24
    print(f"{model} riem n={n:2d} b={b}: {dt * 1e6:9.1f} us  {b / dt / 1e6:9.1f} M pairs/s  {b * bpp / dt / 8e12:.3f} of the HBM roof")
ops.check_status(dev)
