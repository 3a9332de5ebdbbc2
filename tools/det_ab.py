#!/usr/bin/env python3
"""A/B of the two gradient-accumulation forms of the fused training backward:
   atomic   sympa_model_loss_backward: loss + backward + fp64-atomic scatter into the dense gradient, one launch
   rows+seg sympa_model_loss_backward_rows (per-pair rows) + sympa_segment_sum_rows (precomputed order): deterministic
python tools/det_ab.py [model]   -- n = 4 at 65 536 pairs / 5 041 rows and n = 8 at 262 144 pairs / 45 500 rows"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "upper"
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for n, nodes, batch in ((4, 5041, 65536), (6, 5041, 65536), (8, 45500, 262144)):
    table = data.trained_like_table(nodes, n, model=model, seed=1).to(dev)
    trip = data.sample_pairs(nodes, batch, 0, 3).to(dev)
    gd = (1.0 + (trip[:, 0] + trip[:, 1]) % 7).to(torch.float64)
    sc = torch.ones(1, dtype=torch.float64, device=dev)
    gs = torch.zeros(1, dtype=torch.float64, device=dev)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    grad = torch.zeros_like(table)
    rows = torch.empty(2 * batch, 2, n, n, dtype=torch.float64, device=dev)
    order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), nodes)
    t_atomic = timed(lambda: ops.model_loss_backward(table, trip, gd, grad, loss, model, "riem", scale=sc, grad_scale=gs))
    g_atomic = torch.zeros_like(table)
    ops.model_loss_backward(table, trip, gd, g_atomic, loss, model, "riem", scale=sc, grad_scale=gs)
    t_rows = timed(lambda: ops.model_loss_backward_rows(table, trip, gd, rows, loss, model, "riem", scale=sc, grad_scale=gs))
    t_seg = timed(lambda: ops.segment_sum_rows_(grad, rows, order, rowptr))
    ops.check_status(dev)
    err = float((grad - g_atomic).abs().max()) / float(g_atomic.abs().max())
    print(f"{model} n={n} rows={nodes} pairs={batch}: atomic scatter in the kernel {t_atomic:8.1f} us | per-pair rows {t_rows:8.1f} us + "
          f"segmented sum {t_seg:6.1f} us = {t_rows + t_seg:8.1f} us deterministic | max rel diff {err:.1e}", flush=True)
