#!/usr/bin/env python3
"""Soak test of the packed indexed forward (Siegel dims 5..8 both models, spd dims 9..12, 16: ops.SPD_PACKED_DIMS): random table sizes, batch sizes, scales,
metrics, single calls and lists; every result compared with the dense-row kernels of the same library (1e-11 relative) and a 16-pair
sample per case with the CPU oracle (1e-9); poisoned cases (out-of-range index, NaN row, row outside the manifold) must give the
same status bits and NaN positions on both paths.      python tools/fuzz_packed.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import siegel_oracle as orc  # noqa: E402
from sympa_amd import ops  # noqa: E402
from tests.helpers import points, spd_points  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("FUZZ_SEED", "1")))
t0 = time.time()
cases = pairs = poisoned = lists = 0
worst = {"siegel": 0.0, "spd": 0.0, "oracle": 0.0}
METRICS = ("riem", "fone", "finf", "fmin", "wsum")


def status_bits():
    buf = ops._status_buf(dev)
    bits = int(buf.tolist()[0])
    buf.zero_()
    return bits


def rel(a, b):
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), "NaN positions differ"
    if not fin.any():
        return 0.0
    return float(((a[fin] - b[fin]).abs() / b[fin].abs().clamp_min(1e-300)).max())


while time.time() - t0 < budget:
    rows = int(torch.randint(1, 4000, (1,), generator=g))
    b = int(torch.randint(1, 70000, (1,), generator=g)) if torch.rand(1, generator=g) < 0.3 else int(torch.randint(1, 3000, (1,), generator=g))
    s = float(10 ** (-3 * float(torch.rand(1, generator=g))))
    trip = torch.stack((torch.randint(0, rows, (b,), generator=g), torch.randint(0, rows, (b,), generator=g)), 1).to(dev)
    scale = (0.05 + 3 * torch.rand(1, generator=g, dtype=torch.float64)).to(dev)
    poison = torch.rand(1, generator=g) < 0.15
    if torch.rand(1, generator=g) < 0.3:
        n = sorted(ops.SPD_PACKED_DIMS)[int(torch.randint(0, len(ops.SPD_PACKED_DIMS), (1,), generator=g))]    # where the binding packs
        table = spd_points(rows, n, min(s, 0.5), g).to(dev)
        if poison:
            table[int(torch.randint(0, rows, (1,), generator=g))] = float("nan")
        pk = ops.SpdPackedTable().ensure(table)
        a = ops.spd_model_forward_packed(pk, trip, scale, 1.0)
        torch.cuda.synchronize()
        sa = status_bits()
        c = ops.spd_model_forward(table, trip, scale, 1.0) if hasattr(ops, "spd_model_forward") else None
        if c is None:
            c = ops.spd_dist_forward(table[trip[:, 0]], table[trip[:, 1]]) * torch.clamp(scale, min=0.1)
        torch.cuda.synchronize()
        sc = status_bits()
        err = rel(a, c)
        worst["spd"] = max(worst["spd"], err)
        kind = f"spd n={n}"
    else:
        n = int(torch.randint(5, 9, (1,), generator=g))
        model = "upper" if torch.rand(1, generator=g) < 0.5 else "bounded"
        metric = METRICS[int(torch.randint(0, 5, (1,), generator=g))]
        w = torch.rand(n, generator=g, dtype=torch.float64).to(dev) if metric == "wsum" else None
        table = points(model, rows, n, min(s, 0.3), g).to(dev)
        if poison:
            what = int(torch.randint(0, 3, (1,), generator=g))
            if what == 0:
                trip[int(torch.randint(0, b, (1,), generator=g)), int(torch.randint(0, 2, (1,), generator=g))] = rows + 5
            elif what == 1:
                table[int(torch.randint(0, rows, (1,), generator=g))] = float("nan")
            elif model == "upper":
                table[int(torch.randint(0, rows, (1,), generator=g)), 1] *= -1.0          # Im Z negative definite
            else:
                table[int(torch.randint(0, rows, (1,), generator=g))] *= 50.0             # far outside the unit ball
        pk = ops.PackedTable(model).ensure(table)
        if torch.rand(1, generator=g) < 0.3 and b >= 4:
            # the list form: the batch cut into up to 40 pieces of unequal length (more than one launch group)
            k = int(torch.randint(2, 41, (1,), generator=g))
            cuts = sorted(set(int(x) for x in torch.randint(1, b, (k,), generator=g)))
            parts = [trip[i:j].contiguous() for i, j in zip([0] + cuts, cuts + [b])]
            outs = [torch.empty(p.shape[0], dtype=torch.float64, device=dev) for p in parts]
            ops.PackedBatchedForward(pk, table, parts, outs, metric, w, scale, 1.0).run()
            a = torch.cat(outs)
            lists += 1
        else:
            a = ops.model_forward_packed(pk, trip, metric, w, scale, 1.0)
        torch.cuda.synchronize()
        sa = status_bits()
        c = ops.model_forward(table, trip, model, metric, w, scale, 1.0)
        torch.cuda.synchronize()
        sc = status_bits()
        err = rel(a, c)
        worst["siegel"] = max(worst["siegel"], err)
        kind = f"{model} n={n} {metric}"
        if not poison:
            idx = torch.randint(0, b, (min(16, b),), generator=g)
            t_cpu, tr = table.cpu(), trip.cpu()[idx]
            ref = orc.model_forward(t_cpu, tr, model, metric, None if w is None else w.cpu(), scale.cpu(), 1.0)
            # (absolute floor 1e-12: the oracle's d(x, x) is ~1e-15, the packed kernel's exactly 0 -- SURVEY 8d: floor 1e-9)
            e2 = float((((a.cpu()[idx] - ref).abs() - 1e-12).clamp_min(0.0) / ref.abs().clamp_min(1e-9)).max())
            worst["oracle"] = max(worst["oracle"], e2)
            if not e2 < 1e-9:
                print(f"ORACLE MISMATCH {kind} rows={rows} b={b} s={s:.3g} err={e2:.3e}")
                sys.exit(1)
    if sa != sc or (not poison and sa != 0):
        print(f"STATUS MISMATCH {kind} rows={rows} b={b}: packed {sa} dense {sc} poison={bool(poison)}")
        sys.exit(1)
    if not err < 1e-11:
        print(f"MISMATCH {kind} rows={rows} b={b} s={s:.3g} err={err:.3e}")
        sys.exit(1)
    cases += 1
    pairs += b
    poisoned += int(bool(poison))
print(f"fuzz ok: {cases} cases ({poisoned} poisoned, {lists} lists), {pairs} pairs, worst rel diff packed vs dense: siegel {worst['siegel']:.2e}, "
      f"spd {worst['spd']:.2e}; vs the CPU oracle (16 pairs per case) {worst['oracle']:.2e}; {time.time() - t0:.0f} s")
