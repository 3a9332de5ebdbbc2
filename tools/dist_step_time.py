#!/usr/bin/env python3
"""The data-parallel training step (sympa_amd.train_step.DistributedTrainStep: replayed graphs around the gradient exchange)
against the single-GPU two-kernel step, on ONE GPU over RCCL at world size 1 (the N > 1 code path: process group, collectives
on the stream, the flat exchange buffer; what is missing is the xGMI time of the collective itself):
    python tools/dist_step_time.py [steps]
Shapes: the headline (upper / riem / n = 4, 65 536 triplets per step, 5 041 rows) and configs[3] (n = 8, 262 144, 45 500)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sympa_amd import data, ops  # noqa: E402
from sympa_amd.model import Model  # noqa: E402
from sympa_amd.optim import RiemannianSGD  # noqa: E402
from sympa_amd.train_step import DistributedTrainStep, GraphedTrainStep, batches_want_source_order  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
g = torch.Generator().manual_seed(5)


def fresh(manifold, metric, n, nodes):
    class A:
        pass
    A.manifold, A.metric, A.dims, A.num_points = manifold, metric, n, nodes
    A.scale_coef, A.scale_init, A.train_scale = 1.0, 1.0, True
    m = Model(A)
    with torch.no_grad():
        if manifold == "spd":
            m.embeddings.embeds.data = data.spd_table(nodes, n, seed=1)
        else:
            m.embeddings.embeds.data = data.trained_like_table(nodes, n, model=manifold, seed=1)
    return m.to(dev)


def timed(step, trip, batch):
    def run(k):
        step.load_epoch(trip[:batch * k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step.run_steps(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k
    run(3)
    return min(run(steps) for _ in range(3))


for name, manifold, metric, n, nodes, batch in (("headline", "upper", "riem", 4, 5041, 65536),
                                                ("cartesian", "upper", "riem", 8, 45500, 262144),
                                                ("custom-spd", "spd", "riem", 16, 100000, 1048576)):
    if os.environ.get("WORKLOADS") and name not in os.environ["WORKLOADS"].split(","):
        continue
    if name == "custom-spd":
        steps = min(steps, 6)
    trip = torch.stack((torch.randint(0, nodes, (batch * steps,), generator=g), torch.randint(0, nodes, (batch * steps,), generator=g),
                        torch.randint(1, 9, (batch * steps,), generator=g)), 1).to(dev)
    rows = []
    for form in ("single-GPU two-kernel step (atomic scatter)", "single-GPU two-kernel step (deterministic)"):
        m = fresh(manifold, metric, n, nodes)
        opt = RiemannianSGD(m.parameters(), lr=1e-4)
        try:
            st = GraphedTrainStep(m, opt, batch, 50.0, dev, deterministic="determ" in form, accumulate_loss=True)
        except ValueError:
            st = None
        if st is None or st.mode != "two_kernels":
            # dims 7, 8: the classic graph (zero, fused loss + backward, norms, RSGD, scale step), one call per batch
            if "atomic" in form:
                st = GraphedTrainStep(m, opt, batch, 50.0, dev, two_kernels=False)
                first = trip[:batch]
                if batches_want_source_order(m):        # what load_epoch does for the replayed steps below (sympa_amd/data.py)
                    first = data.sort_batches_by_source(first, batch)
                ids, gd = first[:, :2].contiguous(), first[:, 2].to(torch.float64)
                for _ in range(3):
                    st(ids, gd)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    st(ids, gd)
                torch.cuda.synchronize()
                rows.append(("single-GPU classic graph (one call per batch)", (time.perf_counter() - t0) / steps, ""))
            continue
        rows.append((form, timed(st, trip, batch), ""))
    for mode, cap, det in (("dense", True, False), ("dense", False, False), ("dense", True, True), ("rows", False, False),
                           ("sharded", True, False), ("sharded", True, True)):
        if (det and (n > 6 or manifold == "spd")) or (manifold == "spd" and mode == "rows"):
            continue
        m = fresh(manifold, metric, n, nodes)
        opt = RiemannianSGD(m.parameters(), lr=1e-4)
        st = DistributedTrainStep(m, opt, batch, 50.0, dev, mode=mode, capture_collective=cap, deterministic=det)
        st.force_split = not cap            # A/B: backward graph, exchange between the replays, optimiser graph
        dt = timed(st, trip, batch)
        rows.append((f"DistributedTrainStep mode={mode}{', deterministic' if det else ''}, RCCL world 1", dt,
                     f"{st.graphs_per_step} graph(s) per step" + (f" (whole-step capture refused: {st.capture_error[:60]})"
                                                                  if cap and st.graphs_per_step != 1 and hasattr(st, 'capture_error') else "")
                     + f", {st.ex.message_bytes / 1e6:.2f} MB/rank/step at world 8: {int(2 * 7 / 8 * st.ex.flat.numel() * 8) / 1e6:.2f} MB"))
    ops.check_status(dev)
    for form, dt, note in rows:
        print(f"{name:10s} n={n} batch={batch:7d} nodes={nodes:6d}  {form:62s} {dt * 1e6:9.1f} us/step  {note}", flush=True)
dist.barrier()
dist.destroy_process_group()
