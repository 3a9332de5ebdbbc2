#!/usr/bin/env python3
"""Whole training step of the reference's loop (runner.py:98-118: forward, AverageDistortionLoss, backward, gradient clip,
RiemannianSGD step on the table and the scale) as hipGraph replays (sympa_amd/train_step.py), on the BASELINE.json
workloads, every form the step exists in:
   classic         round 2's graph (zero, fused loss+backward, norms, RSGD, scale step), the batch copied in per step
   classic, epoch  the same graph with the batches addressed by the device step counter (round 6: Siegel models), nothing copied per step
   2k, copy        two kernels per step (train_backward + rsgd_step_fused), the batch copied in per step
   2k, epoch       two kernels per step, batches addressed by the device step counter: K replays, nothing in between
   2k, epoch, det  the same with per-pair rows + segmented sum + fixed-order scalar sums (bitwise reproducible)
   python tools/train_step_time.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402
from sympa_amd.optim import RiemannianAdam, RiemannianSGD  # noqa: E402
from sympa_amd.train_step import GraphedTrainStep  # noqa: E402
from tests.helpers import spd_points  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
# (name, manifold, metric, dims, nodes, batch)   -- BASELINE.json configs[0..4]
WORKLOADS = [("grid", "upper", "riem", 2, 125, 512), ("tree", "upper", "riem", 4, 1093, 8192),
             ("margulis", "bounded", "finf", 4, 5041, 65536), ("headline", "upper", "riem", 4, 5041, 65536),
             ("cartesian", "upper", "riem", 8, 45500, 262144), ("custom-spd", "spd", "riem", 16, 100000, 1048576)]
only = os.environ.get("WORKLOADS")
g = torch.Generator().manual_seed(5)


def fresh(manifold, metric, n, nodes):
    class A:
        pass
    A.manifold, A.metric, A.dims, A.num_points = manifold, metric, n, nodes
    A.scale_coef, A.scale_init, A.train_scale = 1.0, 1.0, True
    m = Model(A)
    with torch.no_grad():
        if manifold == "spd":
            m.embeddings.embeds.data = spd_points(nodes, n, 0.3, torch.Generator().manual_seed(5))
        else:
            m.embeddings.embeds.data = data.trained_like_table(nodes, n, model=manifold, seed=1)
    return m.to(dev)


for name, manifold, metric, n, nodes, batch in WORKLOADS:
    if only and name not in only.split(","):
        continue
    k_epoch = max(steps, 8)
    trip = torch.stack((torch.randint(0, nodes, (batch * k_epoch,), generator=g),
                        torch.randint(0, nodes, (batch * k_epoch,), generator=g),
                        torch.randint(1, 9, (batch * k_epoch,), generator=g)), 1).to(dev)
    # the data pipeline's per-epoch sort inside every batch, where the backward's scatter merges equal source rows (sympa_amd/data.py)
    if not os.environ.get("NO_BATCH_SORT") and ((manifold == "upper" and n == 8) or (manifold == "spd" and 9 <= n <= 16)):
        trip = data.sort_batches_by_source(trip, batch)
    ids, gd = trip[:batch, :2].contiguous(), trip[:batch, 2].to(torch.float64)
    for form in ("classic", "classic, epoch", "2k, copy", "2k, epoch", "2k, epoch, det"):
        m = fresh(manifold, metric, n, nodes)
        if os.environ.get("OPTIM", "rsgd") == "radam":           # train.py:69-70
            if manifold == "spd":
                continue
            opt = RiemannianAdam(m.parameters(), lr=1e-4, eps=1e-7, stabilize=None)
        else:
            opt = RiemannianSGD(m.parameters(), lr=1e-4)
        try:
            step = GraphedTrainStep(m, opt, batch, 50.0, dev, two_kernels=not form.startswith("classic"), deterministic=form.endswith("det"),
                                    accumulate_loss=not form.startswith("classic"))
        except ValueError:
            continue
        if not form.startswith("classic") and step.mode != "two_kernels":
            continue
        if form == "classic, epoch" and not step._classic_windowed():
            continue
        if "epoch" in form:
            def run(k):
                step.load_epoch(trip[:batch * k])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step.run_steps(k)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / k
            run(3)
            dt = min(run(k_epoch) for _ in range(3))
        else:
            for _ in range(3):
                step(ids, gd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(ids, gd)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        ops.check_status(dev)
        print(f"{name:10s} {manifold:7s} {metric} n={n:2d} nodes={nodes:6d} batch={batch:7d} {type(opt).__name__[10:]:4s} [{form:14s}]: {dt * 1e6:10.1f} us per "
              f"training step  {batch / dt / 1e6:9.2f} M pairs/s trained", flush=True)
