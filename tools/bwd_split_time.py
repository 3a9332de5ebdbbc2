#!/usr/bin/env python3
"""Where the n = 8 training step goes: dense backward (per-pair rows, no scatter) vs fused step with the in-kernel
scatter vs rows + scatter kernel.   python tools/bwd_split_time.py [n] [b] [nodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 45500
dev = torch.device("cuda:0")
table = data.trained_like_table(nodes, n, seed=1).to(dev)
pairs = data.sample_pairs(nodes, b, 0, 1).to(dev)
gd = torch.rand(b, dtype=torch.float64, device=dev) * 5 + 1
scale = torch.ones(1, dtype=torch.float64, device=dev)
gt = torch.zeros_like(table)
loss = torch.zeros(1, dtype=torch.float64, device=dev)
gs = torch.zeros(1, dtype=torch.float64, device=dev)
rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
idx = torch.cat((pairs[:, 0], pairs[:, 1])).contiguous()
z1, z2 = table[pairs[:, 0]].contiguous(), table[pairs[:, 1]].contiguous()
go = torch.ones(b, dtype=torch.float64, device=dev)


def timeit(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n} b={b}: {name:52s} {dt * 1e6:9.1f} us  {b / dt / 1e6:8.1f} M pairs/s")


timeit("forward (model_forward)", lambda: ops.model_forward(table, pairs, "upper", "riem", None, scale, 1.0))
timeit("dense backward, pre-gathered (siegel_dist_backward)", lambda: ops.siegel_dist_backward(z1, z2, go, "upper", "riem"))
timeit("fused loss + backward, rows out (no scatter)", lambda: ops.model_loss_backward_rows(table, pairs, gd, rows, loss, "upper", "riem", None, None, scale, gs, 1.0, 1.0))
timeit("scatter kernel alone (2b rows)", lambda: ops.scatter_add_rows_(gt, rows.reshape(2 * b, -1), idx))
timeit("fused loss + backward + in-kernel scatter", lambda: ops.model_loss_backward(table, pairs, gd, gt, loss, "upper", "riem", None, None, scale, gs, 1.0, 1.0))
ops.check_status(dev)
