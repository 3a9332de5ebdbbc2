#!/usr/bin/env python3
"""Times the SPD distance kernels: python tools/spd_time.py [n] [batch] [num_rows]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

from sympa_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
a = torch.randn(rows, n, n, generator=g, dtype=torch.float64) * 0.05
a = 0.5 * (a + a.transpose(-1, -2))
table = (torch.eye(n, dtype=torch.float64) + a + a @ a * 0.5).to(dev)
trip = torch.randint(0, rows, (b, 3), generator=g).to(dev)
out = torch.empty(b, dtype=torch.float64, device=dev)
results = {}
for name, fl in (("specialised", 0), ("generic", ops.FLAG_GENERIC)):
    if fl and "--generic" not in sys.argv and n != 16:
        continue
    for _ in range(2):
        ops.spd_model_forward(table, trip, out=out, flags=fl)
    torch.cuda.synchronize()
    reps = 5 if fl else 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spd_model_forward(table, trip, out=out, flags=fl)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / reps
    bytes_per_pair = 2 * n * n * 8 + 16 + 8
    print(f"spd n={n} b={b} rows={rows} {name}: {dt * 1e3:.3f} ms  {b / dt / 1e6:.1f} M pairs/s  "
          f"{b * bytes_per_pair / dt / 1e9:.0f} GB/s algorithmic ({b * bytes_per_pair / dt / 8e12:.3f} of the HBM roof)")
    results[name] = out.clone()
if len(results) == 2:
    a_, b_ = results["specialised"], results["generic"]
    print(f"    max rel diff specialised vs generic: {float(((a_ - b_).abs() / b_.abs().clamp_min(1e-300)).max()):.2e}")

# training path: fused loss + backward rows + scatter, then the RSGD step (one lane per pair / row, scratch)
if "--train" in sys.argv:
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "spd", "riem", n, rows
        scale_coef, scale_init, train_scale = 1.0, 1.0, False
    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = table.cpu()
    m = m.to(dev)
    bt = min(b, 65536)
    gd = torch.randint(1, 9, (bt,), generator=g).to(torch.float64).to(dev)
    m.embeddings.embeds.grad = torch.zeros_like(m.embeddings.embeds.data)
    for _ in range(2):
        m.fused_loss_backward(trip[:bt], gd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.fused_loss_backward(trip[:bt], gd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"spd n={n} fused loss+backward+scatter (Model.fused_loss_backward) b={bt}: {dt * 1e3:.2f} ms  {bt / dt / 1e6:.2f} M pairs/s")
    grad = m.embeddings.embeds.grad
    for _ in range(2):
        ops.spd_rsgd_step_(m.embeddings.embeds.data, grad * 1e-6, 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.spd_rsgd_step_(m.embeddings.embeds.data, grad * 1e-6, 1e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"spd n={n} RSGD step over {rows} rows: {dt * 1e3:.2f} ms")
    ops.check_status(dev)
