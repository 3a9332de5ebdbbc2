#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process on ONE device (cdna guide rule 24): variant i is
selected by the `flags` word of sympa_model_forward; every round replays each variant's 16-launch hipGraph
once; prints median / min us per launch per variant.   usage: python tools/ab_bench.py 0 0x100 [--rounds 200]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data, ops

flags = [int(x, 0) for x in sys.argv[1:] if not x.startswith("--")] or [0, 0]
rounds = 200
dev = torch.device("cuda:0")
nodes, n, batch, nb = 5041, 4, 65536, 16
table = data.trained_like_table(nodes, n).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]
graphs = []
for f in flags:
    for i in range(nb):
        ops.model_forward(table, batches[i], "upper", "riem", None, scale, 1.0, out=outs[i], flags=f)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nb):
            ops.model_forward(table, batches[i], "upper", "riem", None, scale, 1.0, out=outs[i], flags=f)
    graphs.append(g)
times = [[] for _ in flags]
for r in range(rounds + 20):
    for k, g in enumerate(graphs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); g.replay(); b.record()
        torch.cuda.synchronize()
        if r >= 20:
            times[k].append(a.elapsed_time(b) * 1e3 / (2 * nb))
for f, t in zip(flags, times):
    t.sort()
    print(f"flags {f:#x}: median {t[len(t)//2]:.3f} us/launch  min {t[0]:.3f}  p90 {t[int(len(t)*0.9)]:.3f}")
