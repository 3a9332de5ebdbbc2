#!/usr/bin/env python3
"""SPD n = 16 training backward: gradient accumulation by fp64 atomics inside the kernel (sympa_spd_loss_backward) against
per-pair rows + the segmented sum in a precomputed order (sympa_spd_backward_rows + sympa_segment_sum_rows; one stable sort per
batch): python tools/spd_det_ab.py [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
rows_n, n = 100000, 16
dev = torch.device("cuda:0")
table = data.spd_table(rows_n, n, seed=42).to(dev)
trip = data.sample_pairs(rows_n, b, 0, 42).to(dev)
gd = torch.randint(1, 9, (b,), generator=torch.Generator().manual_seed(1)).to(torch.float64).to(dev)
sc = torch.ones(1, dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


grad_a = torch.zeros_like(table)
loss_a = torch.zeros(1, dtype=torch.float64, device=dev)


def atomic():
    grad_a.zero_(); loss_a.zero_()
    ops.spd_loss_backward(table, trip, grad_a, graph_dist=gd, scale=sc, loss=loss_a)


rows = torch.empty(2 * b, n, n, dtype=torch.float64, device=dev)
grad_d = torch.zeros_like(table)
loss_d = torch.zeros(1, dtype=torch.float64, device=dev)
order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), rows_n)


def det(sort=False):
    global order, rowptr
    loss_d.zero_()
    if sort:
        order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), rows_n)
    ops.spd_backward_rows(table, table, trip, graph_dist=gd, scale=sc, loss=loss_d, rows=rows)
    ops.segment_sum_rows_(grad_d, rows.view(2 * b, -1), order[0].contiguous(), rowptr[0].contiguous())


ta, td, tds = timed(atomic), timed(det), timed(lambda: det(True))
big = float(grad_a.abs().max())
print(f"spd n=16 b={b}: atomic scatter in the kernel {ta:.3f} ms | per-pair rows + segmented sum {td:.3f} ms (+ sort per batch: {tds:.3f} ms) | "
      f"max rel diff {float((grad_a - grad_d).abs().max()) / big:.1e}  loss {abs(float(loss_a - loss_d)) / abs(float(loss_a)):.1e}")
ops.check_status(dev)
