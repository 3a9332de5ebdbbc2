#!/usr/bin/env python3
"""Whole training step of the reference's loop (runner.py:98-118: forward, AverageDistortionLoss, backward, gradient clip,
RiemannianSGD step) as ONE hipGraph replay per batch (sympa_amd/train_step.py), on the BASELINE.json workloads.
   python tools/train_step_time.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402
from sympa_amd.optim import RiemannianSGD  # noqa: E402
from sympa_amd.train_step import GraphedTrainStep  # noqa: E402
from tests.helpers import spd_points  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
# (name, manifold, metric, dims, nodes, batch)   -- BASELINE.json configs[0..4]
WORKLOADS = [("grid", "upper", "riem", 2, 125, 512), ("tree", "upper", "riem", 4, 364, 8192),
             ("margulis", "bounded", "finf", 4, 5041, 65536), ("headline", "upper", "riem", 4, 5041, 65536),
             ("cartesian", "upper", "riem", 8, 45500, 262144), ("custom-spd", "spd", "riem", 16, 100000, 1048576)]
g = torch.Generator().manual_seed(5)
for name, manifold, metric, n, nodes, batch in WORKLOADS:
    class A:
        pass
    A.manifold, A.metric, A.dims, A.num_points = manifold, metric, n, nodes
    A.scale_coef, A.scale_init, A.train_scale = 1.0, 1.0, True
    m = Model(A)
    with torch.no_grad():
        if manifold == "spd":
            m.embeddings.embeds.data = spd_points(nodes, n, 0.3, g)
        else:
            m.embeddings.embeds.data = data.trained_like_table(nodes, n, model=manifold, seed=1)
    m = m.to(dev)
    opt = RiemannianSGD(m.parameters(), lr=1e-4)
    step = GraphedTrainStep(m, opt, batch, 50.0, dev)
    ids = torch.stack((torch.randint(0, nodes, (batch,), generator=g), torch.randint(0, nodes, (batch,), generator=g)), 1).to(dev)
    gd = torch.randint(1, 9, (batch,), generator=g).to(torch.float64).to(dev)
    for _ in range(3):
        step(ids, gd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(ids, gd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.check_status(dev)
    print(f"{name:10s} {manifold:7s} {metric} n={n:2d} nodes={nodes:6d} batch={batch:7d}: {dt * 1e6:10.1f} us per training step  "
          f"{batch / dt / 1e6:9.2f} M pairs/s trained")
