#!/usr/bin/env python3
"""Fused training step (loss + backward + scatter) timing by dims: python tools/bwd_time_dims.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402

dev = torch.device("cuda:0")
for n, nodes, b in ((4, 5041, 65536), (6, 5000, 65536), (7, 5000, 65536), (8, 45500, 262144), (10, 5000, 16384), (16, 5000, 16384)):
    table = data.trained_like_table(nodes, n, seed=1).to(dev)
    pairs = data.sample_pairs(nodes, b, 0, 1).to(dev)
    gd = torch.rand(b, dtype=torch.float64, device=dev) * 5 + 1
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    gt = torch.zeros_like(table)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    gs = torch.zeros(1, dtype=torch.float64, device=dev)

    def step():
        gt.zero_()
        return ops.model_loss_backward(table, pairs, gd, gt, loss, "upper", "riem", None, None, scale, gs, 1.0, 1.0)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"fused loss+backward n={n} b={b}: {dt * 1e6:.1f} us  {b / dt / 1e6:.1f} M pairs/s")
