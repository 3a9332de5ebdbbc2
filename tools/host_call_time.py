#!/usr/bin/env python3
"""Host cost of the per-batch calls of the mirrored API (what the reference's unmodified loops issue, runner.py:98-101,126-131):
    python tools/host_call_time.py
For each binding of C-ABI sympa_model_forward -- ctypes (sympa_amd/_lib.py) and the thin torch binding (sympa_amd/_fast) --
  * host time of ONE call (returns when the launch is enqueued; the queue is drained before every sample)
  * steady-state time per call of 2 000 back-to-back calls on one stream (host-bound or kernel-bound, whichever is slower)
for Model.forward(batch) under no_grad (fresh output per call) and ops.model_forward(..., out=) (preallocated output),
headline shape (upper / riem / n = 4 / 65 536 pairs / 5 041 rows); then Model.forward_batches for comparison."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import _lib, data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402

dev = torch.device("cuda:0")


class A:
    manifold, metric, dims, num_points = "upper", "riem", 4, 5041
    scale_coef, scale_init, train_scale = 1.0, 1.0, False


m = Model(A)
with torch.no_grad():
    m.embeddings.embeds.data = data.trained_like_table(5041, 4)
m = m.to(dev)
bl = [data.sample_pairs(5041, 65536, j).to(dev) for j in range(20)]
table, scale = m.embeddings.embeds.data, m.scale.data
outs = [torch.empty(65536, dtype=torch.float64, device=dev) for _ in range(20)]


def measure(name, fn):
    for i in range(50):
        fn(i)
    torch.cuda.synchronize()
    one = []
    for i in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(i)
        one.append(time.perf_counter() - t0)
    one.sort()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        fn(i)
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 2000
    print(f"{name:58s} one call {one[150] * 1e6:6.2f} us (min {one[0] * 1e6:5.2f})   back to back {steady * 1e6:6.2f} us/call = "
          f"{65536 / steady / 1e9:5.2f} G pairs/s")


with torch.no_grad():
    for label, off in (("thin torch binding (_fast)", False), ("ctypes", True)):
        _lib._fast = False if off else None          # False: switched off; None: (re)bind on next use
        if not off and _lib.fast() is None:
            print("sympa_amd._fast is not built")
            continue
        measure(f"Model.forward(batch)                     [{label}]", lambda i: m(bl[i % 20]))
        measure(f"ops.model_forward(table, batch, out=)    [{label}]",
                lambda i: ops.model_forward(table, bl[i % 20], "upper", "riem", None, scale, 1.0, out=outs[i % 20]))
    _lib._fast = None
    plan = m.prepare_batches(bl)
    m.forward_batches(plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        m.forward_batches(plan)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 2000
    print(f"{'Model.forward_batches(plan of 20 batches)':58s} {el * 1e6:6.2f} us/step = {65536 / el / 1e9:5.2f} G pairs/s   (the list-of-batches form)")
ops.check_status(dev)
