#!/usr/bin/env python3
"""Direct multi-launch path (C-ABI sympa_model_forward_batches) against hipGraph replays, K steps of the headline
workload:  python tools/overhead_probe2.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
trials = 100
dev = torch.device("cuda:0")
nodes, n, batch, nb = 5041, 4, 65536, 16
table = data.trained_like_table(nodes, n).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]
ref = [ops.model_forward(table, batches[j], "upper", "riem", None, scale, 1.0).clone() for j in range(nb)]
torch.cuda.synchronize()
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]


def measure(label, flags, streams):
    bf = ops.BatchedForward(table, [batches[i % nb] for i in range(max(K, 128))], [outs[i % nb] for i in range(max(K, 128))],
                            "upper", "riem", None, scale, 1.0, flags=flags, streams=streams)
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        bf.run(0, 128)
        torch.cuda.synchronize()
    for o in outs:
        o.zero_()
    bf.run(0, K)
    torch.cuda.synchronize()
    ok = all(torch.equal(outs[i % nb], ref[i % nb]) for i in range(min(K, nb)))
    ts, hs = [], []
    for t in range(trials + 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bf.run(0, K)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if t >= 10:
            ts.append((t2 - t0) * 1e6); hs.append((t1 - t0) * 1e6)
    ts.sort(); hs.sort()
    print(f"{label:42s} K={K}: wall med {ts[len(ts)//2]:7.1f} min {ts[0]:7.1f} p90 {ts[int(len(ts)*.9)]:7.1f} us "
          f"({ts[len(ts)//2]/K:.2f} us/step), host enqueue med {hs[len(hs)//2]:6.1f} us, outputs_equal={ok}", flush=True)


print(f"HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT')}")
cur = torch.cuda.current_stream(dev)
measure("1 stream, in order, full LDS", 0, [cur])
measure("1 stream, ANY_ORDER, full LDS", ops.FLAG_ANY_ORDER, [cur])
measure("1 stream, ANY_ORDER, low LDS", ops.FLAG_ANY_ORDER | ops.FLAG_LOW_LDS, [cur])
measure("1 side stream, ANY_ORDER, low LDS", ops.FLAG_ANY_ORDER | ops.FLAG_LOW_LDS, pool[:1])
for s in (2, 3, 4, 6, 8):
    measure(f"{s} streams, in order, low LDS", ops.FLAG_LOW_LDS, pool[:s])
measure("4 streams, ANY_ORDER, low LDS", ops.FLAG_ANY_ORDER | ops.FLAG_LOW_LDS, pool[:4])
measure("2 streams, ANY_ORDER, low LDS", ops.FLAG_ANY_ORDER | ops.FLAG_LOW_LDS, pool[:2])
