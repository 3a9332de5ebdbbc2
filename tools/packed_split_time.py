#!/usr/bin/env python3
"""Two-kernel (two lanes per pair) packed forward against the one-kernel packed forward and the dense rows, upper model dims 7, 8.
    python tools/packed_split_time.py [n,nodes,pairs ...]
Prints per shape the time per launch (pair) of: dense, packed one kernel (SYMPA_NO_PACKED_SPLIT path: workspace None), packed split,
and the list form (20 batches) with workspace caps; max rel diff split vs one-kernel and vs dense."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [a.split(",") for a in sys.argv[1:] if not a.startswith("--")] or \
    [(8, 45500, 262144), (7, 45500, 262144), (8, 5041, 262144), (8, 45500, 65536), (8, 45500, 1000003)]
G = 8


def timed(fn, reps=12, g=G):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(g):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / g)
    ts.sort()
    return ts[len(ts) // 2]


for n, nodes, pairs in shapes:
    n, nodes, pairs = int(n), int(nodes), int(pairs)
    table = data.trained_like_table(nodes, n, model="upper", seed=42).to(dev)
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    batches = [data.sample_pairs(nodes, pairs, j, 42).to(dev) for j in range(4)]
    outs = {k: torch.empty(pairs, dtype=torch.float64, device=dev) for k in "dps"}
    pk = ops.PackedTable("upper").ensure(table)
    it = [0]

    def dense():
        it[0] += 1
        ops.model_forward(table, batches[it[0] % 4], "upper", "riem", None, scale, 1.0, out=outs["d"])

    def one():
        it[0] += 1
        os.environ["SYMPA_NO_PACKED_SPLIT"] = "1"
        ops.model_forward_packed(pk, batches[it[0] % 4], "riem", None, scale, 1.0, out=outs["p"])

    def split():
        it[0] += 1
        os.environ.pop("SYMPA_NO_PACKED_SPLIT", None)
        ops.model_forward_packed(pk, batches[it[0] % 4], "riem", None, scale, 1.0, out=outs["s"])

    for f in (dense, one, split):
        it[0] = 0
        f()
    torch.cuda.synchronize()
    ops.check_status(dev)
    rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1e-300)).max())
    e_sp, e_sd = rel(outs["s"], outs["p"]), rel(outs["s"], outs["d"])
    for metric in ("fone", "finf", "fmin", "wsum"):
        w = torch.rand(n, dtype=torch.float64, device=dev) if metric == "wsum" else None
        os.environ["SYMPA_NO_PACKED_SPLIT"] = "1"
        a = ops.model_forward_packed(pk, batches[0], metric, w, scale, 1.0)
        os.environ.pop("SYMPA_NO_PACKED_SPLIT", None)
        b = ops.model_forward_packed(pk, batches[0], metric, w, scale, 1.0)
        e_sp = max(e_sp, rel(b, a))
    t_d, t_p, t_s = timed(dense), timed(one), timed(split)
    bpp = 32 * n * n + 24
    print(f"upper n={n} N={nodes:6d} b={pairs:7d}  dense {t_d:8.1f}  packed-1k {t_p:8.1f}  split {t_s:8.1f} us "
          f"(frac {pairs * bpp / (t_s * 1e-6) / 8e12:.3f})   split vs 1k {e_sp:.2e}  vs dense {e_sd:.2e}", flush=True)
    if pairs > 300000:
        continue
    # list form: 20 batches, workspace capped at `cap` pairs
    K = 20
    lb = [data.sample_pairs(nodes, pairs, 100 + j, 42).to(dev) for j in range(K)]
    lo = [torch.empty(pairs, dtype=torch.float64, device=dev) for _ in range(K)]
    ref = None
    for cap in (0, 32768, 65536, 131072, 262144, 1048576):
        if cap == 0:
            os.environ["SYMPA_NO_PACKED_SPLIT"] = "1"
        else:
            os.environ.pop("SYMPA_NO_PACKED_SPLIT", None)
            ops.PACKED_SPLIT_LIST_PAIRS = cap
            pk._ws = None
        bf = ops.PackedBatchedForward(pk, table, lb, lo, "riem", None, scale, 1.0)
        bf.run()
        torch.cuda.synchronize()
        cat = torch.cat(lo)
        if ref is None:
            ref = cat.clone()
        t = timed(bf.run, reps=8, g=2)
        print(f"    list of {K}: workspace cap {cap:8d} pairs  {t / K:8.1f} us per batch   max rel diff vs one-kernel {rel(cat, ref):.2e}", flush=True)
