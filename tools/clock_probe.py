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
NB = 64


def stamp():
    buf = torch.zeros(3 * NB, dtype=torch.int64, device=dev)
    _lib.check(lib.sympa_clock_stamp(buf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return buf


def clocks(a, b):
    a, b = a.cpu().view(NB, 3), b.cpu().view(NB, 3)
    out = {}
    for i in range(NB):
        if a[i, 2] != b[i, 2]:
            continue
        dt, dr = int(b[i, 0] - a[i, 0]), int(b[i, 1] - a[i, 1])
        if dr > 0:
            out.setdefault(int(a[i, 2]), []).append(dt / dr * 100.0)
    return {x: sum(v) / len(v) for x, v in sorted(out.items())}, (int(b[0, 1] - a[0, 1]) / 100.0)


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
                res.append(clocks(s0, s1))
            for c, us in res:
                vals = list(c.values())
                print(f"{model} n={n} [{label}] region {us:9.1f} us  clock MHz per XCD: " + " ".join(f"{x}:{v:6.0f}" for x, v in c.items()) +
                      f"   mean {sum(vals) / max(len(vals), 1):6.0f}", flush=True)
