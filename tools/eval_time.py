#!/usr/bin/env python3
"""Evaluation epoch of the harness (Runner.evaluate, runner.py:124-135) on configs[1]'s graph: balanced tree b = 3, h = 6,
596 778 (i < j, d) triplets, batch 8192, upper / riem / n = 4 -- Model.evaluate (one C call, fused multi-batch launches)
against one Model.forward call per batch (round 2's harness)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402

dev = torch.device("cuda:0")
trip, id2node = data.graph_triplets(data.named_graph("tree-b3-h6"))


class A:
    manifold, metric, dims, num_points = "upper", "riem", 4, len(id2node)
    scale_coef, scale_init, train_scale = 1.0, 1.0, False


m = Model(A)
with torch.no_grad():
    m.embeddings.embeds.data = data.trained_like_table(len(id2node), 4, seed=1)
m = m.to(dev)
ids, gd = trip[:, :2].contiguous().to(dev), trip[:, 2].to(torch.float64).to(dev)
batch = 8192


def loop():
    tot = torch.zeros(1, dtype=torch.float64, device=dev)
    with torch.no_grad():
        for s in range(0, ids.shape[0], batch):
            d = m(ids[s:s + batch])
            tot += ((d - gd[s:s + batch]).abs() / gd[s:s + batch]).sum()
    return float(tot) / ids.shape[0]


for name, fn in (("Model.evaluate (fused multi-batch kernel, one C call)", lambda: m.evaluate(ids, gd, batch)),
                 ("one Model.forward per batch (round 2's harness)", loop)):
    for _ in range(3):
        v = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        v = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e3:7.3f} ms per evaluation epoch of {ids.shape[0]} triplets ({ids.shape[0] / dt / 1e9:.2f} G triplets/s), "
          f"average distortion {v:.6f}")
ops.check_status(dev)
