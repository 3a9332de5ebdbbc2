#!/usr/bin/env python3
"""Soak test of the SPLIT Siegel backward (dims 5..8, both models, SYMPA_FLAG_SPLIT: two kernels through a workspace) against the
one-stage one-lane kernels (SYMPA_FLAG_GENERIC): random dims, batch sizes (ragged waves included), scales and metrics, some pairs
identical, the per-pair rows form and the fused scatter form (random and source-sorted batches: the merged rows of the n = 8
scatter).  Round 6: in a third of the cases pairs with GRADED spectra (eigenvalues of E^H E spread over 1e-6 .. 1e-12) are planted at
random positions, so that some waves take the hand-over to the one-stage kernel (stage 1's flag words + siegel_bwd_list_kernel) and
their neighbours do not.   python tools/fuzz_split_bwd.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import ops  # noqa: E402
from tests.helpers import graded_pairs, points, to_bounded  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("FUZZ_SEED", "1")))
t0 = time.time()
cases = pairs = graded_cases = 0
worst_rows = worst_table = worst_loss = p99 = 0.0


def rowerr(got, ref, b):
    scale = ref.abs().reshape(b, -1).max(1).values.clamp_min(1e-300)
    return (got - ref).abs().reshape(b, -1).max(1).values / scale


while time.time() - t0 < budget:
    n = int(torch.randint(5, 9, (1,), generator=g))
    model = "upper" if torch.rand(1, generator=g) < 0.6 else "bounded"
    metric = ("riem", "fone", "finf", "fmin", "wsum")[int(torch.randint(0, 5, (1,), generator=g))]
    s = min(float(10 ** (-3 * float(torch.rand(1, generator=g)))), 0.4)        # 1e-3 .. 0.4
    w = torch.rand(n, generator=g, dtype=torch.float64)
    if torch.rand(1, generator=g) < 0.5:
        b = int(torch.randint(1, 6000, (1,), generator=g))
        go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
        z1, z2 = points(model, b, n, s, g), points(model, b, n, s, g)
        if b > 3:
            z2[1] = z1[1]
        if torch.rand(1, generator=g) < 0.33 and b > 8:
            k = int(torch.randint(1, min(b, 40), (1,), generator=g))
            ga, gb = graded_pairs(k, n, int(torch.randint(3, 7, (1,), generator=g)), seed=int(torch.randint(0, 10 ** 6, (1,), generator=g)))
            ga, gb = torch.from_numpy(ga), torch.from_numpy(gb)
            if model == "bounded":
                ga, gb = to_bounded(ga), to_bounded(gb)
            where = torch.randperm(b, generator=g)[:k]
            z1[where], z2[where] = ga, gb
            graded_cases += 1
        z1, z2 = z1.to(dev), z2.to(dev)
        a = ops.siegel_dist_backward(z1, z2, go, model=model, metric=metric, weights=w, flags=ops.FLAG_SPLIT)
        c = ops.siegel_dist_backward(z1, z2, go, model=model, metric=metric, weights=w, flags=ops.FLAG_GENERIC)
        err = torch.cat((rowerr(a[0], c[0], b), rowerr(a[1], c[1], b)))
        worst_rows = max(worst_rows, float(err.quantile(0.999)) if metric != "riem" else float(err.max()))
        p99 = max(p99, float(err.quantile(0.99)))
    else:
        nodes = int(torch.randint(20, 3000, (1,), generator=g))
        b = int(torch.randint(64, 9000, (1,), generator=g))
        table = points(model, nodes, n, s, g)
        trip = torch.stack((torch.randint(0, nodes, (b,), generator=g), torch.randint(0, nodes, (b,), generator=g)), 1)
        if torch.rand(1, generator=g) < 0.33 and nodes >= 80:
            k = int(torch.randint(1, 30, (1,), generator=g))
            ga, gb = graded_pairs(k, n, int(torch.randint(3, 7, (1,), generator=g)), seed=int(torch.randint(0, 10 ** 6, (1,), generator=g)))
            ga, gb = torch.from_numpy(ga), torch.from_numpy(gb)
            if model == "bounded":
                ga, gb = to_bounded(ga), to_bounded(gb)
            table[:k], table[k:2 * k] = ga, gb                 # rows 0..k-1 / k..2k-1: the two ends of the graded pairs
            where = torch.randperm(b, generator=g)[:k]
            trip[where, 0], trip[where, 1] = torch.arange(k), torch.arange(k) + k
            graded_cases += 1
        table = table.to(dev)
        if torch.rand(1, generator=g) < 0.5:
            trip = trip[torch.argsort(trip[:, 0], stable=True)]
        trip = trip.contiguous().to(dev)
        gd = (torch.rand(b, generator=g, dtype=torch.float64) * 5 + 1).to(dev)
        sc = torch.full((1,), 1.2, dtype=torch.float64, device=dev)
        res = []
        for flags in (ops.FLAG_SPLIT, ops.FLAG_GENERIC):
            gt = torch.zeros_like(table)
            loss = torch.zeros(1, dtype=torch.float64, device=dev)
            gs = torch.zeros(1, dtype=torch.float64, device=dev)
            ops.model_loss_backward(table, trip, gd, gt, loss, model, metric, w.to(dev), None, sc, gs, 1.0, 1.0, flags=flags)
            res.append((gt, loss, gs))
        big = float(res[1][0].abs().max())
        worst_table = max(worst_table, float((res[0][0] - res[1][0]).abs().max()) / max(big, 1e-300))
        worst_loss = max(worst_loss, abs(float(res[0][1] - res[1][1])) / max(abs(float(res[1][1])), 1e-300))
    ops._status_buf(dev).zero_()
    cases += 1
    pairs += b
print(f"{cases} cases ({graded_cases} with graded pairs planted), {pairs} pairs in {time.time() - t0:.0f} s: split vs one-stage kernels -- per-pair rows worst {worst_rows:.2e} "
      f"(p99 {p99:.2e}; riem: max, the other metrics: 99.9th percentile -- equal eigenvalues leave their subgradient to the basis), "
      f"table gradient worst {worst_table:.2e} of its largest entry, loss worst {worst_loss:.2e}")
