#!/usr/bin/env python3
"""n = 4 backward (per-pair rows form, upper model, 5 041 rows) at growing batch sizes: one wave per SIMD is all a 65 536-pair batch
offers, so the two-waves-per-SIMD build (round 6: 256 registers) can only show from 131 072 pairs on.  Run once per library
(SYMPA_HIP_LIB=build_ab/old_n4bwd.so for the 294-register build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

dev = torch.device("cuda:0")
table = data.trained_like_table(5041, 4, model="upper", seed=1).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
loss = torch.zeros(1, dtype=torch.float64, device=dev)
for b in (65536, 131072, 262144, 1048576):
    trip = data.sample_pairs(5041, b, 0, 3).to(dev)
    gd = (1.0 + (trip[:, 0] + trip[:, 1]) % 7).to(torch.float64)
    rows = torch.empty(2 * b, 2, 4, 4, dtype=torch.float64, device=dev)
    grad = torch.zeros_like(table)

    def run_rows():
        ops.model_loss_backward_rows(table, trip, gd, rows, loss, "upper", "riem", scale=scale)

    def run_atomic():
        ops.model_loss_backward(table, trip, gd, grad, loss, "upper", "riem", scale=scale)

    res = []
    for fn in (run_rows, run_atomic):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8):
                fn()
            c.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(c) * 1e3 / 8)
        ts.sort()
        res.append(ts[len(ts) // 2])
    ops.check_status(dev)
    print(f"upper n=4 b={b:8d}: rows {res[0]:8.1f} us ({b / res[0]:7.1f} M pairs/s)   atomic scatter {res[1]:8.1f} us", flush=True)
