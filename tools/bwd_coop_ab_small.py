#!/usr/bin/env python3
"""A/B at n = 5..8: one pair per lane (registers + scratch) against eight lanes per pair, two pairs per DPP row
(SYMPA_FLAG_COOP) -- dense rows (sympa_siegel_dist_bwd) and the fused loss + backward + scatter step.
   python tools/bwd_coop_ab_small.py [batch] [model]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sympa_amd import ops  # noqa: E402
from tests.helpers import points  # noqa: E402

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
model = sys.argv[2] if len(sys.argv) > 2 else "upper"
g = torch.Generator().manual_seed(7)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3


for n in (5, 6, 7, 8):
    z1, z2 = points(model, b, n, 0.3, g).to(dev), points(model, b, n, 0.3, g).to(dev)
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    rows_n = 5041
    table = points(model, rows_n, n, 0.3, g).to(dev)
    trip = torch.stack((torch.randint(0, rows_n, (b,), generator=g), torch.randint(0, rows_n, (b,), generator=g)), 1).to(dev)
    gd = torch.randint(1, 9, (b,), generator=g).to(torch.float64).to(dev)
    res = {}
    for name, fl in (("one pair per lane", ops.FLAG_GENERIC), ("eight lanes per pair", ops.FLAG_COOP)):
        dt = timed(lambda: ops.siegel_dist_backward(z1, z2, go, model=model, flags=fl))
        out = ops.siegel_dist_backward(z1, z2, go, model=model, flags=fl)
        gt = torch.zeros_like(table)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        df = timed(lambda: ops.model_loss_backward(table, trip, gd, gt, loss, model=model, flags=fl))
        gt.zero_(); loss.zero_()
        ops.model_loss_backward(table, trip, gd, gt, loss, model=model, flags=fl)
        res[name] = (dt, df, out, gt.clone(), float(loss))
    ops.check_status(dev)
    ref, got = res["one pair per lane"], res["eight lanes per pair"]
    diff = max(float((got[2][k] - ref[2][k]).abs().max() / ref[2][k].abs().max()) for k in (0, 1))
    dgt = float((got[3] - ref[3]).abs().max() / ref[3].abs().max())
    for name, (dt, df, _, _, ls) in res.items():
        print(f"{model} n={n} b={b} {name:22s}: dense rows {dt * 1e6:8.1f} us   fused loss+backward+scatter {df * 1e6:8.1f} us   "
              f"(max rel diff rows {diff:.1e}, table gradient {dgt:.1e}, loss {ls:.6e})")
