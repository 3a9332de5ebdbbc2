#!/usr/bin/env python3
"""Where the fixed cost of a SHORT timed region goes (bench.py --steps 20 is ~100 us of GPU work):
    python tools/overhead_probe.py [K] [--trials 200]
For a K-launch hipGraph of the headline workload on 1/2/4 streams, prints the wall time of
(replay + wait) with the wait done as torch.cuda.synchronize() or as an event-query spin, beside the
device time of the same replay (HIP events).  Run it again under HSA_ENABLE_INTERRUPT=0 to see the
effect of polling signal waits in the runtime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 20
trials = 200
dev = torch.device("cuda:0")
nodes, n, batch, nb = 5041, 4, 65536, 16
table = data.trained_like_table(nodes, n).to(dev)
scale = torch.ones(1, dtype=torch.float64, device=dev)
batches = [data.sample_pairs(nodes, batch, j).to(dev) for j in range(nb)]
outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]


def step(i, flags):
    ops.model_forward(table, batches[i % nb], "upper", "riem", None, scale, 1.0, out=outs[i % nb], flags=flags)


def capture(k, streams, flags):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream(device=dev) for _ in range(streams - 1)]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(main)
        for i in range(k):
            if i % streams == 0:
                step(i, flags)
            else:
                with torch.cuda.stream(side[i % streams - 1]):
                    step(i, flags)
        for s in side:
            main.wait_stream(s)
    return g


for i in range(nb):
    step(i, 0)
torch.cuda.synchronize()
print(f"K={K} HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT')} "
      f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
for streams, flags in ((1, 0), (2, 1), (4, 1), (4, 0), (3, 1)):
    g = capture(K, streams, flags)
    big = capture(128, streams, flags)
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        big.replay()
        torch.cuda.synchronize()
    res = {}
    for mode in ("sync", "spin", "device"):
        ts = []
        for t in range(trials + 10):
            g.replay()                      # keep the clock up between trials
            torch.cuda.synchronize()
            if mode == "device":
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record()
                torch.cuda.synchronize()
                dt = a.elapsed_time(b) * 1e-3
            else:
                ev = torch.cuda.Event()
                t0 = time.perf_counter()
                g.replay()
                if mode == "spin":
                    ev.record()
                    while not ev.query():
                        pass
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            if t >= 10:
                ts.append(dt * 1e6)
        ts.sort()
        res[mode] = (ts[len(ts) // 2], ts[0], ts[int(len(ts) * 0.9)])
    # sustained: 4096 steps
    reps = 4096 // 128
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        big.replay()
    torch.cuda.synchronize()
    sus = (time.perf_counter() - t0) / (reps * 128) * 1e6
    line = "  ".join(f"{m}: med {v[0]:7.1f} min {v[1]:7.1f} p90 {v[2]:7.1f}" for m, v in res.items())
    print(f"streams={streams} flags={flags}:  {line}  | per step (sync med) {res['sync'][0]/K:.2f} us, "
          f"device {res['device'][0]/K:.2f} us, sustained(4096) {sus:.2f} us")
