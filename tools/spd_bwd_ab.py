#!/usr/bin/env python3
"""A/B of the spd backward kernels: one lane per pair (SYMPA_FLAG_GENERIC) against sixteen lanes per pair at M = n.
   python tools/spd_bwd_ab.py [batch] [dims ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sympa_amd import ops  # noqa: E402
from tests.helpers import spd_points  # noqa: E402

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dims = [int(a) for a in sys.argv[2:]] or list(range(6, 17))
g = torch.Generator().manual_seed(7)
for n in dims:
    rows_n = 4096
    table = spd_points(rows_n, n, 0.3, g).to(dev)
    trip = torch.stack((torch.randint(0, rows_n, (b,), generator=g), torch.randint(0, rows_n, (b,), generator=g)), 1).to(dev)
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    rows = torch.empty(2 * b, n, n, dtype=torch.float64, device=dev)
    res = {}
    for name, fl in (("one lane per pair", ops.FLAG_GENERIC), ("sixteen lanes per pair", ops.FLAG_COOP), ("sixteen lanes, paired QL", 0)):
        for _ in range(2):
            ops.spd_backward_rows(table, table, trip, grad_out=go, rows=rows, flags=fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.spd_backward_rows(table, table, trip, grad_out=go, rows=rows, flags=fl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res[name] = (dt, rows.clone())
    ops.check_status(dev)
    ref = res["one lane per pair"][1]
    for name, (dt, rows_) in res.items():
        diff = float((rows_ - ref).abs().max() / ref.abs().max())
        print(f"spd backward rows n={n:2d} b={b} {name:24s}: {dt * 1e6:9.1f} us  {b / dt / 1e6:8.2f} M pairs/s   max rel diff {diff:.1e}")
