#!/usr/bin/env python3
"""Interleaved A/B timing of two (or more) BUILDS of libsympa_hip.so in one process on one device:
    python tools/ab_lib.py build_ab/base.so build_ab/new.so [--rounds 200]
Each variant gets its own ctypes handle and its own 128-launch hipGraph of the headline workload; every round
replays each graph once; prints median / min / p90 us per launch and checks that the outputs agree."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data

paths = [a for a in sys.argv[1:] if not a.startswith("--")]
streams = 4 if "--overlap" in sys.argv else 1       # --overlap: launches alternate over 4 streams, minimum-LDS kernel form
flags = 1 if streams > 1 else 0
rounds = 200
dev = torch.device("cuda:0")
nodes, n, batch, nb, gn = 5041, 4, 65536, 16, 128
table = data.trained_like_table(nodes, n).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
status = torch.zeros(2, dtype=torch.int32, device=dev)
variants = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    fn = lib.sympa_model_forward
    fn.restype = ctypes.c_int
    V, I64, I, D = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    fn.argtypes = [V, I64, I, V, I64, V, I64, I64, I, I, V, D, V, D, V, V, I, V]
    outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]

    def launch(i, fn=fn, outs=outs):
        t = batches[i % nb]
        rc = fn(table.data_ptr(), nodes, n, t.data_ptr(), 2, t.data_ptr() + 8, 2, batch, 0, 0, None, 1e-5,
                scale.data_ptr(), 1.0, outs[i % nb].data_ptr(), status.data_ptr(), flags,
                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    for i in range(nb):
        launch(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream(device=dev) for _ in range(streams - 1)]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for st_ in side:
            st_.wait_stream(main)
        for i in range(gn):
            if i % streams == 0:
                launch(i)
            else:
                with torch.cuda.stream(side[i % streams - 1]):
                    launch(i)
        for st_ in side:
            main.wait_stream(st_)
    variants.append((p, g, outs))
times = [[] for _ in variants]
for r in range(rounds + 20):
    for k, (_, g, _) in enumerate(variants):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        if r >= 20:
            times[k].append(a.elapsed_time(b) * 1e3 / gn)
ref = variants[0][2]
for (p, _, outs), t in zip(variants, times):
    t.sort()
    same = all(torch.allclose(o, r_, rtol=1e-12, atol=1e-14) for o, r_ in zip(outs, ref))
    print(f"{p}: median {t[len(t)//2]:.3f} us/launch  min {t[0]:.3f}  p90 {t[int(len(t)*0.9)]:.3f}  outputs_match={same}")
