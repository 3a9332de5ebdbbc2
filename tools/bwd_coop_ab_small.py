#!/usr/bin/env python3
"""A/B at n = 7, 8 (upper, dense rows): one pair per lane (registers + scratch) against sixteen lanes per pair (SYMPA_FLAG_COOP)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sympa_amd import ops  # noqa: E402
from tests.helpers import points  # noqa: E402

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
g = torch.Generator().manual_seed(7)
for n in (7, 8):
    z1, z2 = points("upper", b, n, 0.3, g).to(dev), points("upper", b, n, 0.3, g).to(dev)
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    res = {}
    for name, fl in (("one pair per lane", 0), ("sixteen lanes per pair", ops.FLAG_COOP)):
        for _ in range(2):
            out = ops.siegel_dist_backward(z1, z2, go, flags=fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = ops.siegel_dist_backward(z1, z2, go, flags=fl)
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / 3, out)
    ops.check_status(dev)
    ref, got = res["one pair per lane"][1], res["sixteen lanes per pair"][1]
    diff = max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in (0, 1))
    for name, (dt, _) in res.items():
        print(f"upper backward n={n} b={b} {name:24s}: {dt * 1e6:9.1f} us  {b / dt / 1e6:8.2f} M pairs/s   max rel diff {diff:.1e}")
