#!/usr/bin/env python3
"""Split backward (two kernels, one pair per lane, dims 5..8) against the one-launch kernels, fused loss + backward + scatter:
    python tools/bwd_split_ab.py [--dims 8] [--models upper,bounded] [--rows]
prints us per call (HIP events over 10 calls, gradient zeroing excluded) and the largest relative difference of the table gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops  # noqa: E402

dims = [int(x) for x in (sys.argv[sys.argv.index("--dims") + 1].split(",") if "--dims" in sys.argv else "5,6,7,8".split(","))]
models = sys.argv[sys.argv.index("--models") + 1].split(",") if "--models" in sys.argv else ["upper", "bounded"]
rows_form = "--rows" in sys.argv
dev = torch.device("cuda:0")
for model in models:
    for n in dims:
        nodes, b = (45500, 262144) if n == 8 else (5000, 65536)
        table = data.trained_like_table(nodes, n, seed=1, model=model).to(dev)
        pairs = data.sample_pairs(nodes, b, 0, 1).to(dev)
        if "--sorted" in sys.argv:              # the batch sorted by its first column (what sympa_amd/train_step.py loads)
            pairs = pairs[torch.argsort(pairs[:, 0], stable=True)].contiguous()
        gd = torch.rand(b, dtype=torch.float64, device=dev) * 5 + 1
        scale = torch.ones(1, dtype=torch.float64, device=dev)
        res = {}
        if "--split-only" in sys.argv:          # timing of a variant build (SYMPA_HIP_LIB=build_ab/x.so): no comparison
            gt = torch.zeros_like(table)
            loss = torch.zeros(1, dtype=torch.float64, device=dev)
            gs = torch.zeros(1, dtype=torch.float64, device=dev)
            ws = torch.empty(ops.siegel_backward_workspace_bytes(b, n, model), dtype=torch.uint8, device=dev)
            step = lambda: ops.model_loss_backward(table, pairs, gd, gt, loss, model, "riem", None, None, scale, gs, 1.0, 1.0, flags=ops.FLAG_SPLIT, workspace=ws)
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            torch.cuda.synchronize()
            print(f"{os.environ.get('SYMPA_HIP_LIB', 'product')}: {model} n={n} b={b} split {e0.elapsed_time(e1) * 100:.1f} us", flush=True)
            ops._status_buf(dev).zero_()
            continue
        for name, flags in (("split", ops.FLAG_SPLIT), ("coop", ops.FLAG_COOP), ("one-lane", ops.FLAG_GENERIC)):
            gt = torch.zeros_like(table)
            rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev) if rows_form else None
            loss = torch.zeros(1, dtype=torch.float64, device=dev)
            gs = torch.zeros(1, dtype=torch.float64, device=dev)
            ws = torch.empty(max(ops.siegel_backward_workspace_bytes(b, n, model), 16), dtype=torch.uint8, device=dev)

            def step():
                if rows_form:
                    return ops.model_loss_backward_rows(table, pairs, gd, rows, loss, model, "riem", None, None, scale, gs, 1.0, 1.0,
                                                        flags=flags, workspace=ws if flags == ops.FLAG_SPLIT else None)
                return ops.model_loss_backward(table, pairs, gd, gt, loss, model, "riem", None, None, scale, gs, 1.0, 1.0,
                                               flags=flags, workspace=ws if flags == ops.FLAG_SPLIT else None)

            step()
            torch.cuda.synchronize()
            res[name] = (rows if rows_form else gt).clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[name + "_us"] = e0.elapsed_time(e1) * 100
        ops._status_buf(dev).zero_()
        den = res["coop"].abs().max().item()
        print(f"{model} n={n} b={b} {'rows' if rows_form else 'scatter'}: split {res['split_us']:.1f} us | eight lanes {res['coop_us']:.1f} us | "
              f"one lane one kernel {res['one-lane_us']:.1f} us | split vs eight lanes rel {(res['split'] - res['coop']).abs().max().item() / den:.2e}"
              f"  one-lane vs eight lanes {(res['one-lane'] - res['coop']).abs().max().item() / den:.2e}", flush=True)
