#!/usr/bin/env python3
"""Time of the optimiser-side row operations:  python tools/table_time.py [model] [rows] [dims ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sympa_amd import ops  # noqa: E402
from tests.helpers import points  # noqa: E402

dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "upper"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5041
dims = [int(a) for a in sys.argv[3:]] or [4, 8, 10, 16]
g = torch.Generator().manual_seed(3)
for n in dims:
    table = points(model, rows, n, 0.3, g).to(dev)
    grad = torch.randn(rows, 2, n, n, generator=g, dtype=torch.float64)
    grad = (0.5 * (grad + grad.transpose(-1, -2))).to(dev)

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    t_e = timed(lambda: ops.egrad2rgrad(table, grad, model))
    tab = table.clone()
    t_s = timed(lambda: ops.rsgd_step_(tab, grad, model, 1e-6, 0.0))
    t_p = timed(lambda: ops.projx(table, model))
    print(f"{model} n={n:2d} rows={rows}: egrad2rgrad {t_e:8.1f} us   rsgd step {t_s:8.1f} us   projx {t_p:8.1f} us")
    ops.check_status(dev)
