#!/usr/bin/env python3
"""spd forward: the dense kernel (factorisation per pair) against the packed one (sympa_spd_table_pack once per table version).
    python tools/spd_packed_time.py [n,nodes,pairs ...]      default: configs[4] (16, 100 000, 1 048 576) and n = 6..15"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

ops.SPD_PACKED_DIMS = frozenset(range(6, 17))            # (the binding's default restricts the packed path to the dims where it is the faster one)
dev = torch.device("cuda:0")
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or \
    [(16, 100000, 1048576)] + [(n, 100000, 262144) for n in (15, 14, 13, 12, 11, 10, 9, 8, 6)]
G = 4


def timed(fn, reps=10):
    for _ in range(2):
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


for n, nodes, pairs in shapes:
    table = data.spd_table(nodes, n, seed=42).to(dev)
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    batches = [data.sample_pairs(nodes, pairs, j, 42).to(dev) for j in range(2)]
    out_d = torch.empty(pairs, dtype=torch.float64, device=dev)
    out_p = torch.empty(pairs, dtype=torch.float64, device=dev)
    pk = ops.SpdPackedTable().ensure(table)
    it = [0]

    def dense():
        it[0] += 1
        ops.spd_model_forward(table, batches[it[0] % 2], scale, 1.0, out=out_d)

    def packed():
        it[0] += 1
        ops.spd_model_forward_packed(pk, batches[it[0] % 2], scale, 1.0, out=out_p)

    def repack():
        pk.invalidate()
        pk.ensure(table)

    it[0] = 0
    dense()
    it[0] = 0
    packed()
    torch.cuda.synchronize()
    ops.check_status(dev)
    err = float(((out_p - out_d).abs() / out_d.abs().clamp_min(1e-300)).max())
    t_d, t_p, t_k = timed(dense), timed(packed), timed(repack)
    bpp = 16 * n * n + 24
    print(f"spd n={n:2d} N={nodes} b={pairs:7d}  dense {t_d:8.1f} us (frac {pairs * bpp / (t_d * 1e-6) / 8e12:.3f})   packed {t_p:8.1f} us "
          f"(frac {pairs * bpp / (t_p * 1e-6) / 8e12:.3f})   pack alone {t_k:7.1f} us   max rel diff {err:.2e}", flush=True)
