#!/usr/bin/env python3
"""Indexed forward, dims 5..8: the one-kernel forward over the reference's dense rows against the two-kernel forward over the
packed table (ops.PackedTable, C-ABI sympa_table_pack / sympa_model_forward_packed), same box, interleaved.
    python tools/packed_fwd_time.py [model,n,nodes,pairs ...]      default: the configs[3] shape + dims 5..8 of both models
Prints per shape: pack time, dense / packed time per launch (HIP events around groups of sequential launches), pack + packed (the
table changes every step), the contract fraction (32 n^2 + 24 B per pair against 8 TB/s) and max rel diff packed vs dense."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [a.split(",") for a in sys.argv[1:] if not a.startswith("--")] or \
    [("upper", 8, 45500, 262144), ("upper", 7, 45500, 262144), ("upper", 6, 45500, 262144), ("upper", 5, 45500, 262144),
     ("bounded", 8, 45500, 262144), ("bounded", 7, 45500, 262144), ("bounded", 6, 45500, 262144), ("bounded", 5, 45500, 262144),
     ("upper", 8, 5041, 262144), ("upper", 8, 45500, 65536)]
G = 8


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(G):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / G)
    ts.sort()
    return ts[len(ts) // 2]


for model, n, nodes, pairs in shapes:
    n, nodes, pairs = int(n), int(nodes), int(pairs)
    table = data.trained_like_table(nodes, n, model=model, seed=42).to(dev)
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    batches = [data.sample_pairs(nodes, pairs, j, 42).to(dev) for j in range(4)]
    out_d = torch.empty(pairs, dtype=torch.float64, device=dev)
    out_p = torch.empty(pairs, dtype=torch.float64, device=dev)
    pk = ops.PackedTable(model).ensure(table)
    it = [0]

    def dense():
        it[0] += 1
        ops.model_forward(table, batches[it[0] % 4], model, "riem", None, scale, 1.0, out=out_d)

    def packed():
        it[0] += 1
        ops.model_forward_packed(pk, batches[it[0] % 4], "riem", None, scale, 1.0, out=out_p)

    def repack():
        pk.invalidate()
        pk.ensure(table)

    def repack_and_packed():
        repack()
        packed()

    it[0] = 0
    dense()
    it[0] = 0
    packed()
    torch.cuda.synchronize()
    ops.check_status(dev)
    err = float(((out_p - out_d).abs() / out_d.abs().clamp_min(1e-300)).max())
    t_d, t_p, t_k, t_kp = timed(dense), timed(packed), timed(repack), timed(repack_and_packed)
    bpp = 32 * n * n + 24
    print(f"{model:8s} n={n} N={nodes:6d} b={pairs:7d}  dense {t_d:8.1f} us (frac {pairs * bpp / (t_d * 1e-6) / 8e12:.3f})   packed {t_p:8.1f} us "
          f"(frac {pairs * bpp / (t_p * 1e-6) / 8e12:.3f})   pack alone {t_k:7.1f} us   pack + packed {t_kp:8.1f} us   "
          f"max rel diff {err:.2e}", flush=True)
