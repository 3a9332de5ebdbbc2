#!/usr/bin/env python3
"""SPD backward (n = 16 by default): the three-kernel form (eigenvectors one pair per lane by inverse iteration, vectors parked in a
caller-owned workspace; csrc/spd_coop_bwd3_kernel.hpp) against the kernel that runs the QL with accumulated rotations in the
sixteen-lanes layout -- same inputs, gradients compared, both timed.
    python tools/spd_bwd3_ab.py [pairs] [rows] [n = 9..16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
table = data.spd_table(rows, n, seed=42).to(dev)
trip = data.sample_pairs(rows, b, 0, 42).to(dev)
if os.environ.get("SORTED_BATCH"):           # the batch sorted by its first column (sympa_amd/data.py::sort_batches_by_source)
    trip = data.sort_batches_by_source(trip, b)
g = torch.Generator().manual_seed(1)
gd = torch.randint(1, 9, (b,), generator=g).to(torch.float64).to(dev)
sc = torch.ones(1, dtype=torch.float64, device=dev)


def run(flags, reps=5):
    # a persistent workspace, as the replayed training step holds one (sympa_amd/train_step.py): a fresh 3.4 GB tensor per call
    # makes the timing depend on the caching allocator's mood (7.6 .. 29 ms measured for the same kernels)
    need = 0 if os.environ.get("SYMPA_SPD_BWD_NO_WORKSPACE") else ops._lib.load().sympa_spd_backward_workspace_bytes(b, n)
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if need > 0 else None
    grad = torch.zeros_like(table)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    gs = torch.zeros(1, dtype=torch.float64, device=dev)
    out = ops.spd_loss_backward(table, trip, grad, graph_dist=gd, scale=sc, loss=loss, grad_scale=gs, want_out=True, flags=flags,
                                workspace=ws)
    ops.check_status(dev)
    torch.cuda.synchronize()
    scratch = torch.zeros_like(table)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spd_loss_backward(table, trip, scratch, graph_dist=gd, scale=sc, loss=loss.clone(), flags=flags, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    return grad, loss, gs, out, e0.elapsed_time(e1) / reps


new = run(0)
os.environ["SYMPA_SPD_BWD_NO_WORKSPACE"] = "1"
old = run(0)
del os.environ["SYMPA_SPD_BWD_NO_WORKSPACE"]
big = float(old[0].abs().max())
print(f"spd n={n} b={b} rows={rows}: three-phase {new[4]:.3f} ms   QL-with-vectors (two rounds together) {old[4]:.3f} ms   "
      f"x{old[4] / new[4]:.2f}")
print(f"  grad max abs diff / max {float((new[0] - old[0]).abs().max()) / big:.2e}   loss rel diff "
      f"{abs(float(new[1] - old[1])) / abs(float(old[1])):.2e}   dist max rel diff "
      f"{float(((new[3] - old[3]).abs() / old[3].abs().clamp_min(1e-300)).max()):.2e}   grad_scale rel diff "
      f"{abs(float(new[2] - old[2])) / max(1e-300, abs(float(old[2]))):.2e}")
