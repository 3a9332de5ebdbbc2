#!/usr/bin/env python3
"""configs[3] backward (upper n = 8, 262 144 pairs of 45 500 rows, batch sorted by source row): the split backward with the ATOMIC
scatter inside its gradient kernel against the split backward writing PER-PAIR ROWS + the deterministic segmented sum
(sympa_segment_sum_rows: the form the dims <= 6 training step already takes from 32 768 pairs on).  HIP events, median of 8.
    python tools/n8_rows_vs_atomics.py [n] [pairs] [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 45500
dev = torch.device("cuda:0")
table = data.trained_like_table(rows_n, n, model="upper", seed=1).to(dev)
g = torch.Generator().manual_seed(5)
trip = torch.stack((torch.randint(0, rows_n, (b,), generator=g), torch.randint(0, rows_n, (b,), generator=g),
                    torch.randint(1, 9, (b,), generator=g)), 1).to(dev)
trip = data.sort_batches_by_source(trip, b)
gd = trip[:, 2].to(torch.float64).contiguous()
ws = ops.siegel_backward_workspace(b, n, "upper", dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
grad_a = torch.zeros_like(table)
grad_r = torch.zeros_like(table)
rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
loss = torch.zeros(1, dtype=torch.float64, device=dev)
order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), rows_n)


def atomic():
    ops.model_loss_backward(table, trip, gd, grad_a, loss, "upper", "riem", scale=scale, workspace=ws)


def rows_only():
    ops.model_loss_backward_rows(table, trip, gd, rows, loss, "upper", "riem", scale=scale, workspace=ws)


def seg_only():
    ops.segment_sum_rows_(grad_r, rows, order, rowptr)


def rows_seg():
    rows_only()
    seg_only()


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        c.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(c) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


grad_a.zero_()
atomic()
rows_seg()
torch.cuda.synchronize()
ops.check_status(dev)
err = float((grad_a - grad_r).abs().max() / grad_a.abs().max())
print(f"upper n={n} b={b} rows={rows_n}: atomic scatter {timed(atomic):8.1f} us   rows {timed(rows_only):8.1f} us + segmented sum {timed(seg_only):8.1f} us "
      f"= {timed(rows_seg):8.1f} us   max |diff| / max {err:.2e}", flush=True)
