#!/usr/bin/env python3
"""The shader clock the chip holds under the forward kernels (C-ABI sympa_clock_stamp around back-to-back launches).
    python tools/clock_probe.py
For each workload: stamp, ~40 ms of back-to-back launches of the list-form forward, stamp -> MHz per XCD; then the same around ONE
K = 20 region after an idle gap (what the driver's timed region sees)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import _lib, data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import clock_util  # noqa: E402


def stamp():
    return clock_util.stamp(lib, dev, torch)


for model, metric, n, nodes, batch in (("upper", "riem", 4, 5041, 65536), ("upper", "riem", 8, 45500, 262144)):
    class A:
        manifold, dims, num_points = model, n, nodes
        scale_coef, scale_init, train_scale = 1.0, 1.0, False
    A.metric = metric
    net = Model(A)
    with torch.no_grad():
        net.embeddings.embeds.data = data.trained_like_table(nodes, n, model=model, seed=42)
    net = net.to(dev)
    batches = [data.sample_pairs(nodes, batch, j, 42).to(dev) for j in range(4)]
    plan = net.prepare_batches([batches[i % 4] for i in range(20)])
    with torch.no_grad():
        for _ in range(20):
            net.forward_batches(plan)
        torch.cuda.synchronize()
        for label, reps, idle in (("sustained", 400 if n == 4 else 20, 0.0), ("one K=20 region after 50 ms idle", 1, 0.05),
                                  ("one K=20 region after 1 s idle", 1, 1.0)):
            res = []
            for _ in range(3):
                time.sleep(idle)
                s0 = stamp()
                for _ in range(reps):
                    net.forward_batches(plan)
                s1 = stamp()
                torch.cuda.synchronize()
                res.append(clock_util.between(s0, s1))
            for c in res:
                print(f"{model} n={n} [{label}] region {c['region_us']:9.1f} us  clock {c['mhz']:6.0f} MHz (median of {c['cus_paired']} CUs; "
                      f"min {c['mhz_min_cu']:6.0f} max {c['mhz_max_cu']:6.0f})  per XCD: " +
                      " ".join(f"{x}:{v:5.0f}" for x, v in c['mhz_per_xcd'].items()), flush=True)
