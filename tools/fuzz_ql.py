#!/usr/bin/env python3
"""Soak test of the n = 5..8 kernels (Householder + lockstep QL, where the number of sweeps a pair gets depends on its
wave): GPU results against the g++ build of the same per-pair arithmetic (tests/hostsim), random scales and metrics.
    python tools/fuzz_ql.py [pairs per case]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import ops  # noqa: E402
from tests.helpers import METRICS, hostsim_dist, points  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
worst = 0.0
t0 = time.time()
for n in (5, 6, 7, 8):
    for model in ("upper", "bounded"):
        for s in (1e-3, 0.1, 0.5, 1.0):
            metric = METRICS[int(torch.randint(0, 5, (1,), generator=g))]
            w = torch.rand(n, generator=g, dtype=torch.float64)
            z1, z2 = points(model, b, n, s, g), points(model, b, n, s, g)
            got = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric, w.to(dev)).cpu().numpy()
            ops.check_status(dev)
            want, _, st = hostsim_dist(z1.numpy(), z2.numpy(), model, metric, w.numpy())
            assert st == 0
            err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
            worst = max(worst, err)
            print(f"n={n} {model:7s} s={s:<6g} {metric}: max rel diff {err:.2e}", flush=True)
            if not err < 1e-9:
                sys.exit(1)
print(f"fuzz_ql ok: worst {worst:.2e}, {time.time() - t0:.0f} s")
