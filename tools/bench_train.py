#!/usr/bin/env python3
"""Secondary measurement (not the BASELINE metric): one training step of the reference's loop
(runner.py:98-105) = Model.forward -> AverageDistortionLoss -> backward, on the bench workload.
Reports us per step for forward only, backward kernel only, and the full autograd step."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402
from sympa_amd.losses import AverageDistortionLoss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--nodes", type=int, default=5041)
ap.add_argument("--dims", type=int, default=4)
ap.add_argument("--model", default="upper")
ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
table = data.trained_like_table(a.nodes, a.dims, model=a.model).to(dev)
pairs = [data.sample_pairs(a.nodes, a.batch, j).to(dev) for j in range(8)]
gd = torch.randint(1, 12, (a.batch,), device=dev).to(torch.float64)
go = torch.rand(a.batch, device=dev, dtype=torch.float64)
scale = torch.ones(1, device=dev, dtype=torch.float64)


def timed(fn, n):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


grad = torch.zeros_like(table)
fwd = timed(lambda i: ops.model_forward(table, pairs[i % 8], a.model, "riem", None, scale, 1.0), a.steps)
bwd = timed(lambda i: ops.model_backward(table, pairs[i % 8], go, a.model, "riem", None, scale, 1.0, grad_table=grad), a.steps)
t = table.clone().requires_grad_(True)
loss_fn = AverageDistortionLoss()
from sympa_amd import autograd as sa  # noqa: E402


def full(i):
    t.grad = None
    out = sa.model_forward(t, pairs[i % 8], a.model, "riem", None, scale, 1.0)
    loss_fn.calculate_loss(gd, out).backward()


step = timed(full, a.steps)
loss = torch.zeros(1, device=dev, dtype=torch.float64)
fused = timed(lambda i: (grad.zero_(), ops.model_loss_backward(table, pairs[i % 8], gd, grad, loss, a.model, "riem",
                                                              scale=scale)), a.steps)
print(f"fused training step (zero grad + one kernel: fwd + loss + bwd + scatter): {fused:.1f} us "
      f"-> {a.batch / fused:.3g} M pairs/s trained")
print(f"{a.model} n={a.dims} batch={a.batch}: forward {fwd:.1f} us  backward kernel {bwd:.1f} us  "
      f"autograd step (fwd + loss + bwd, direct launches) {step:.1f} us  -> {a.batch / step:.3g} M pairs/s trained")
