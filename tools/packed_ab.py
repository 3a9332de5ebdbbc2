#!/usr/bin/env python3
"""Interleaved A/B timing of sympa_model_forward_packed across BUILDS of the library (tools/build_variant.sh):
    python tools/packed_ab.py <model> <dims> <nodes> <pairs> a.so b.so ...
HIP events around groups of 8 sequential calls; outputs compared with the first build's."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data

args = sys.argv[1:]
model, n, nodes, batch = args[0], int(args[1]), int(args[2]), int(args[3])
paths = args[4:]
dev = torch.device("cuda:0")
mid = {"upper": 0, "bounded": 1}[model]
table = data.trained_like_table(nodes, n, model=model).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
nb = 4
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
status = torch.zeros(2, dtype=torch.int32, device=dev)
variants = []
V, I64, I, D = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    lib.sympa_table_pack_bytes.restype = I64
    lib.sympa_table_pack_bytes.argtypes = [I64, I, I]
    lib.sympa_table_pack.argtypes = [V, I64, I, I, V, I64, V, V]
    fn = lib.sympa_model_forward_packed
    fn.restype = I
    fn.argtypes = [V, I64, I64, I, V, I64, V, I64, I64, I, I, V, D, V, D, V, V, I, V]
    pb = lib.sympa_table_pack_bytes(nodes, n, mid)
    pack = torch.empty(pb, dtype=torch.uint8, device=dev)
    assert lib.sympa_table_pack(table.data_ptr(), nodes, n, mid, pack.data_ptr(), pb, status.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]

    def launch(i, fn=fn, outs=outs, pack=pack, pb=pb):
        t = batches[i % nb]
        rc = fn(pack.data_ptr(), pb, nodes, n, t.data_ptr(), 2, t.data_ptr() + 8, 2, batch, mid, 0, None, 1e-5,
                scale.data_ptr(), 1.0, outs[i % nb].data_ptr(), status.data_ptr(), 0,
                torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    for i in range(nb):
        launch(i)
    torch.cuda.synchronize()
    variants.append((p, launch, outs))
times = [[] for _ in variants]
G = 8
for r in range(25):
    for k, (_, launch, _) in enumerate(variants):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(G):
            launch(i)
        b.record()
        torch.cuda.synchronize()
        if r >= 5:
            times[k].append(a.elapsed_time(b) * 1e3 / G)
ref = variants[0][2]
bpp = 32 * n * n + 24
for (p, _, outs), t in zip(variants, times):
    t.sort()
    err = max(float(((o - r_).abs() / r_.abs().clamp_min(1e-300)).max()) for o, r_ in zip(outs, ref))
    med = t[len(t) // 2]
    print(f"{model} n={n} N={nodes} b={batch} {os.path.basename(p):28s} median {med:9.2f} us/call  min {t[0]:9.2f}  "
          f"frac {batch * bpp / (med * 1e-6) / 8e12:.3f}  max rel diff vs first {err:.2e}")
