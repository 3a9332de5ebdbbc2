#!/usr/bin/env python3
"""Probe for the unexplained wrong-result builds of the n = 16 sixteen-lanes backward kernels (docs/DESIGN_rounds_3_4.md section 13):
   SYMPA_HIP_LIB=build_ab/<variant>.so python tools/miscompile_probe.py [n] [model]
runs the rows-out backward and the fused scatter form of that build against the one-lane-per-pair kernel of the same
build and prints the worst relative row error (1e-10 = fine, O(1) = the miscompilation)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import _lib, ops, selfcheck  # noqa: E402
from tests.helpers import points  # noqa: E402

selfcheck.ENABLED = False          # the gate would route a wrong kernel away: here we want to see it
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = sys.argv[2] if len(sys.argv) > 2 else "upper"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
worst = 0.0
for b in (1024, 999, 4096 + 7):
    z1, z2 = points(model, b, n, 0.3, g).to(dev), points(model, b, n, 0.3, g).to(dev)
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    a = ops.siegel_dist_backward(z1, z2, go, model=model)
    c = ops.siegel_dist_backward(z1, z2, go, model=model, flags=ops.FLAG_GENERIC)
    for side, (x, y) in enumerate(zip(a[:2], c[:2])):
        scale = y.abs().reshape(b, -1).max(1).values.clamp_min(1e-300)
        err = (x - y).abs().reshape(b, -1).max(1).values / scale
        bad = (err > 1e-6).nonzero().flatten().tolist()
        worst = max(worst, float(err.max()))
        # which pairs: position inside the wave (64 pairs per wave, pair 4 t + g sits in group g, round t) and element
        where = collections.Counter((i % 64) for i in bad)
        el = (x - y).abs().reshape(b, -1)
        cols = collections.Counter(int(el[i].argmax()) for i in bad[:200])
        nn = n * n
        re_err = float((el[:, :nn].max(1).values / scale).max())
        im_err = float((el[:, nn:].max(1).values / scale).max())
        print(f"  b={b} grad_z{side + 1}: worst error of the Re plane {re_err:.2e}, of the Im plane {im_err:.2e}")
        print(f"  b={b} grad_z{side + 1}: {len(bad)} of {b} pairs wrong (> 1e-6); first {bad[:8]}; last-wave pairs wrong: "
              f"{sum(1 for i in bad if i >= b - b % 64)}; by position in the wave (top 6) {where.most_common(6)}; "
              f"worst element index (top 4) {cols.most_common(4)}")
b = 999
z1, z2 = points(model, b, n, 0.3, g).to(dev), points(model, b, n, 0.3, g).to(dev)
table = torch.cat((z1, z2))[:500].contiguous()
trip = torch.randint(0, 500, (b, 2), generator=g).to(dev)
gd = torch.ones(b, dtype=torch.float64, device=dev)
ga, gc = torch.zeros_like(table), torch.zeros_like(table)
la, lc = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
ops.model_loss_backward(table, trip, gd, ga, la, model, "riem")
ops.model_loss_backward(table, trip, gd, gc, lc, model, "riem", flags=ops.FLAG_GENERIC)
scat = float((ga - gc).abs().max() / gc.abs().max())
print(f"{os.path.basename(_lib.LIB_PATH)} {model} n={n}: rows-out backward worst rel row error {worst:.2e}; fused scatter form {scat:.2e}; "
      f"loss {abs(float(la - lc)) / abs(float(lc)):.1e}")
