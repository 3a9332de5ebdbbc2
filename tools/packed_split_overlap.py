#!/usr/bin/env python3
"""Experiment: the two-kernel packed forward, list form, on TWO streams with a workspace each (front of one group overlaps the eigen
kernel of the other).  python tools/packed_split_overlap.py [n nodes pairs K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import data, ops  # noqa: E402

dev = torch.device("cuda:0")
n, nodes, pairs, K = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (8, 45500, 262144, 20)))
table = data.trained_like_table(nodes, n, model="upper", seed=42).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
pk = ops.PackedTable("upper").ensure(table)
lb = [data.sample_pairs(nodes, pairs, 100 + j, 42).to(dev) for j in range(K)]
lo = [torch.empty(pairs, dtype=torch.float64, device=dev) for _ in range(K)]
need = pairs * (n * n + 1) * 8 + 64 * (n * n + 1) * 8


def timed(fn, reps=8, g=2):
    for _ in range(2):
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


os.environ["SYMPA_NO_PACKED_SPLIT"] = "1"
one = ops.PackedBatchedForward(pk, table, lb, lo, "riem", None, scale, 1.0)
one.run()
torch.cuda.synchronize()
ref = torch.cat(lo).clone()
print(f"one kernel, one stream: {timed(one.run) / K:8.1f} us per batch", flush=True)
os.environ.pop("SYMPA_NO_PACKED_SPLIT")
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    parts = []
    for s in range(S):
        bf = ops.PackedBatchedForward(pk, table, lb[s::S], lo[s::S], "riem", None, scale, 1.0)
        bf.workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        parts.append(bf)
    ev = [torch.cuda.Event() for _ in range(S + 1)]

    def run():
        cur = torch.cuda.current_stream(dev)
        ev[S].record(cur)
        for s in range(S):
            streams[s].wait_event(ev[S])
            with torch.cuda.stream(streams[s]):
                parts[s].run()
                ev[s].record(streams[s])
        for s in range(S):
            cur.wait_event(ev[s])

    run()
    torch.cuda.synchronize()
    err = float(((torch.cat(lo) - ref).abs() / ref.abs().clamp_min(1e-300)).max())
    print(f"split, {S} stream(s):     {timed(run) / K:8.1f} us per batch   max rel diff {err:.2e}", flush=True)
