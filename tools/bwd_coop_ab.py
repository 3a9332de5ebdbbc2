#!/usr/bin/env python3
"""A/B of the Siegel backward for dims 9..16: one lane per pair over scratch (SYMPA_FLAG_GENERIC) against sixteen lanes
per pair.   python tools/bwd_coop_ab.py [model] [batch] [dims ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sympa_amd import ops  # noqa: E402
from tests.helpers import points  # noqa: E402

dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "upper"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dims = [int(a) for a in sys.argv[3:]] or [10, 16]
g = torch.Generator().manual_seed(7)
for n in dims:
    z1, z2 = points(model, b, n, 0.3, g), points(model, b, n, 0.3, g)
    z1, z2 = z1.to(dev), z2.to(dev)
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    res = {}
    for name, fl in (("one lane per pair", ops.FLAG_GENERIC), ("sixteen lanes per pair", 0)):
        for _ in range(2):
            out = ops.siegel_dist_backward(z1, z2, go, model=model, flags=fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = ops.siegel_dist_backward(z1, z2, go, model=model, flags=fl)
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / 3, out)
    ops.check_status(dev)
    ref = res["one lane per pair"][1]
    got = res["sixteen lanes per pair"][1]
    diff = max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in (0, 1))
    for name, (dt, _) in res.items():
        print(f"{model} backward n={n:2d} b={b} {name:24s}: {dt * 1e6:9.1f} us  {b / dt / 1e6:8.2f} M pairs/s   max rel diff {diff:.1e}")
