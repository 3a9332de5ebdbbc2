#!/usr/bin/env python3
"""Soak test of the sixteen-lanes-per-pair kernels (inline-asm DPP): random dims, batch sizes and scales, every result
compared with the runtime-n one-lane-per-pair kernel (FLAG_GENERIC).   python tools/fuzz_coop.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import ops  # noqa: E402
from tests.helpers import points, spd_points  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("FUZZ_SEED", "1")))
t0 = time.time()
cases = worst_spd = worst_sg = 0
pairs = 0
while time.time() - t0 < budget:
    b = int(torch.randint(1, 5000, (1,), generator=g))
    s = float(10 ** (-3 * float(torch.rand(1, generator=g))))        # 1e-3 .. 1
    if torch.rand(1, generator=g) < 0.4:
        n = int(torch.randint(6, 17, (1,), generator=g))
        s = min(s, 0.5)
        x, y = spd_points(b, n, s, g).to(dev), spd_points(b, n, s, g).to(dev)
        a = ops.spd_dist_forward(x, y)
        c = ops.spd_dist_forward(x, y, flags=ops.FLAG_GENERIC)
        err = float(((a - c).abs() / c.abs().clamp_min(1e-300)).max())
        worst_spd = max(worst_spd, err)
        lim = 1e-9
    else:
        n = int(torch.randint(9, 17, (1,), generator=g))
        model = "upper" if torch.rand(1, generator=g) < 0.5 else "bounded"
        metric = ("riem", "fone", "finf", "fmin", "wsum")[int(torch.randint(0, 5, (1,), generator=g))]
        s = min(s, 0.3)
        w = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
        z1, z2 = points(model, b, n, s, g).to(dev), points(model, b, n, s, g).to(dev)
        a = ops.siegel_dist_forward(z1, z2, model, metric, w)
        c = ops.siegel_dist_forward(z1, z2, model, metric, w, flags=ops.FLAG_GENERIC)
        err = float(((a - c).abs() / c.abs().clamp_min(1e-300)).max())
        worst_sg = max(worst_sg, err)
        lim = 1e-8
    ops.check_status(dev)
    cases += 1
    pairs += b
    if not err < lim:
        print(f"MISMATCH n={n} b={b} s={s:.3g} err={err:.3e}")
        sys.exit(1)
print(f"fuzz ok: {cases} cases, {pairs} pairs, worst rel diff spd {worst_spd:.2e}, siegel {worst_sg:.2e}, {time.time() - t0:.0f} s")
