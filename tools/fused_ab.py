#!/usr/bin/env python3
"""Interleaved A/B of the headline path (sympa_model_forward_batches + SYMPA_FLAG_FUSE: K batches per call, up to 32 per
launch) across BUILDS of the library on one device:   python tools/fused_ab.py a.so b.so [--k 20]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data

paths = [a for a in sys.argv[1:] if not a.startswith("--") and a.endswith(".so")]
K = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 20
model = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "upper"
mid = {"upper": 0, "bounded": 1}[model]
dev = torch.device("cuda:0")
nodes, n, batch, nb = 5041, 4, 65536, 16
table = data.trained_like_table(nodes, n, model=model).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
status = torch.zeros(4, dtype=torch.int32, device=dev)
V, I64, I, D = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
variants = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    fn = lib.sympa_model_forward_batches
    fn.restype = I
    fn.argtypes = [V, I64, I, V, I64, V, I, I, I, V, D, V, D, V, V, I, V, I]
    outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]
    trip = (V * K)(*[batches[i % nb].data_ptr() for i in range(K)])
    bb = (I64 * K)(*[batch] * K)
    oo = (V * K)(*[outs[i % nb].data_ptr() for i in range(K)])
    st = (V * 1)(torch.cuda.current_stream().cuda_stream)

    def run(fn=fn, trip=trip, bb=bb, oo=oo, st=st):
        rc = fn(table.data_ptr(), nodes, n, trip, 2, bb, K, mid, 0, None, 1e-5, scale.data_ptr(), 1.0, oo, status.data_ptr(), 8, st, 1)
        assert rc == 0, rc
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    variants.append((p, run, outs))
times = [[] for _ in variants]
for r in range(60):
    for k, (_, run, _) in enumerate(variants):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            run()
        b.record()
        torch.cuda.synchronize()
        if r >= 10:
            times[k].append(a.elapsed_time(b) * 1e3 / (4 * K))
for (p, _, outs), t in zip(variants, times):
    t.sort()
    same = all(torch.equal(o, r_) for o, r_ in zip(outs, variants[0][2]))
    med = t[len(t) // 2]
    print(f"{os.path.basename(p):24s} {model} K={K}: median {med:.3f} us/step  min {t[0]:.3f}  p90 {t[int(len(t) * 0.9)]:.3f}  "
          f"{batch / med / 1e3:.2f} G pairs/s  bit-identical to first: {same}")
