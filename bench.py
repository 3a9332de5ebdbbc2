#!/usr/bin/env python3
"""bench.py -- pairwise Siegel distances/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (fused Model.forward: gather + distance + metric + scale,
C-ABI `sympa_model_forward`) over one batch of synthetic pairs, inputs already resident in HBM.
Default workload: upper / riem / n=4, batch 65 536 pairs per GPU, table of 5 041 nodes
(the configuration the headline target is quoted on).  Weak scaling: every rank
processes its own 65 536-pair shard of a (65 536 x N)-pair global batch (DistributedSampler-style
interleave rank::N), table replicated, NO data-path collective.

Prints ONE JSON line on rank 0 (see DESIGN.md section 7 for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (model, metric, dims, nodes, batch per GPU)  -- BASELINE.json configs
    "upper-riem-n4-b65536": ("upper", "riem", 4, 5041, 65536),     # headline
    "tree-upper-riem-n4-b8192": ("upper", "riem", 4, 1093, 8192),  # configs[1]
    "grid-upper-riem-n2-b512": ("upper", "riem", 2, 125, 512),     # configs[0]
    "margulis-bounded-finf-n4-b65536": ("bounded", "finf", 4, 5041, 65536),  # configs[2]
    "cartesian-upper-riem-n8-b262144": ("upper", "riem", 8, 45500, 262144),  # configs[3], per-GPU shard 32768 at 8 GPUs
    "custom-spd-n16-b1048576": ("spd", "riem", 16, 100000, 1048576),         # configs[4] (parity unpinned: geoopt absent)
}

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MEASURED_COPY_GBS = 6290.0   # same guide, chip table: measured copy bandwidth
FP64_VALU_PEAK_TFLOPS = 78.6   # AMD datasheet (vector fp64); reported as the honest second roof


def algorithmic_bytes_per_pair(n, model="upper"):
    """2 int64 indices + two fp64 points ([2,n,n] Siegel, [n,n] spd) + one fp64 output, no reuse credit."""
    planes = 1 if model == "spd" else 2
    return 2 * 8 + 2 * (planes * n * n * 8) + 8


def cpu_baseline(model, metric, n, nodes, batch, seed, budget_s=12.0):
    """Oracle (op-for-op torch-CPU fp64 restatement of the reference) timed on the host cores."""
    import torch
    from oracle import siegel_oracle as so
    from sympa_amd import data
    if model == "spd":
        table = data.spd_table(nodes, n, seed=seed)
        batch = min(batch, 65536)
        one = torch.ones(1, dtype=torch.float64)

        def oracle_forward(tab, pr, _model, _metric):
            return so.spd_model_forward(tab, pr, one, 1.0)
    else:
        table = data.trained_like_table(nodes, n, model=model, seed=seed)
        oracle_forward = so.model_forward
    pairs = data.sample_pairs(nodes, batch, 0, seed)
    # batched LAPACK eigh on tiny matrices does not scale with threads and collapses
    # when oversubscribed: pick the fastest of a few thread counts on a small probe, report that one
    ncpu = os.cpu_count() or 1
    probe = pairs[: min(batch, 8192)]
    best = (None, 0.0)
    with torch.no_grad():
        for threads in sorted({1, min(8, ncpu), min(32, ncpu), ncpu}):
            torch.set_num_threads(threads)
            oracle_forward(table, probe[:1024], model, metric)     # warm
            t0 = time.perf_counter()
            oracle_forward(table, probe, model, metric)
            rate = len(probe) / (time.perf_counter() - t0)
            if rate > best[1]:
                best = (threads, rate)
    torch.set_num_threads(best[0])
    with torch.no_grad():
        done, t0 = 0, time.perf_counter()
        iters = 0
        while True:
            oracle_forward(table, pairs, model, metric)
            done += batch
            iters += 1
            el = time.perf_counter() - t0
            if el > budget_s or iters >= 50:
                break
    return {"value": done / el, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} x batch {batch} of the same workload, oracle/siegel_oracle.py "
                      f"{'spd_model_forward' if model == 'spd' else 'model_forward'}, "
                      f"{el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--workload", default="upper-riem-n4-b65536", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override pairs per GPU per step")
    ap.add_argument("--table", default="trained", choices=["trained", "init"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-batches", type=int, default=16)
    ap.add_argument("--graph-nodes", type=int, default=128,
                    help="kernel launches (= steps) captured per hipGraph; one replay costs ~10 us of host/launch "
                         "overhead whatever its length")
    ap.add_argument("--streams", type=int, default=4,
                    help="graph launch only: the steps of the timed region are captured on this many parallel HIP "
                         "streams, so that independent steps (different batches, different outputs) overlap on the "
                         "GPU: the next steps gather while the previous ones compute (minimum-LDS kernel form, three "
                         "blocks per CU); 1 = strictly sequential launches")
    ap.add_argument("--launch", default="graph", choices=["graph", "direct"],
                    help="graph: the steps are replayed from a captured hipGraph of --distinct-batches kernel "
                         "nodes (one node = one step); direct: one Python->C-ABI call per step")
    args = ap.parse_args()

    import torch
    # stdout carries exactly ONE line, the JSON record: everything else that writes to fd 1 -- RCCL prints its version
    # banner and its warnings there from native code -- is sent to stderr for the lifetime of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    from sympa_amd import _lib, data, ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    _lib.load()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SYMPA_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, barriers, max-over-ranks) on one GPU
    use_dist = world > 1 or bool(os.environ.get("SYMPA_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    model, metric, n, nodes, batch = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    if model == "spd":
        table_cpu = data.spd_table(nodes, n, seed=args.seed)
    else:
        table_cpu = (data.trained_like_table(nodes, n, model=model, seed=args.seed) if args.table == "trained"
                     else data.init_table(nodes, n, seed=args.seed))
    if args.table == "init" and model == "bounded":
        raise SystemExit("init table is defined for the upper model")
    table = table_cpu.to(dev)
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    # global batch j has batch*world pairs; this rank takes the interleave rank::world (weak scaling)
    nb = max(1, min(args.distinct_batches, args.steps))
    batches = []
    for j in range(nb):
        glob = data.sample_pairs(nodes, batch * world, j, args.seed)
        batches.append(glob[rank::world].contiguous().to(dev))
    outs = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]
    out = outs[0]
    if args.launch == "direct":
        args.streams = 1
    flags = ops.FLAG_LOW_LDS if (args.streams > 1 and not os.environ.get('SYMPA_BENCH_FULL_LDS')) else 0
    flags |= int(os.environ.get('SYMPA_BENCH_FLAGS', '0'), 0)

    def step(i, fl=None):
        if model == "spd":
            ops.spd_model_forward(table, batches[i % nb], scale, 1.0, out=outs[i % nb])
            return
        ops.model_forward(table, batches[i % nb], model, metric, None, scale, 1.0, out=outs[i % nb],
                          flags=flags if fl is None else fl)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- launch plan: K steps = K kernel launches, either direct or replayed from hipGraphs whose nodes are
    # the launches themselves (a step is still exactly one kernel over one batch).  Two graphs are captured:
    # a long one (--graph-nodes launches) and a short one (one cycle of the nb distinct batches); K steps =
    # as many long replays as fit, then short ones, then direct launches for the last < nb steps.
    def capture(nodes, streams=None, fl=None):
        streams = args.streams if streams is None else streams
        g_ = torch.cuda.CUDAGraph()
        side = [torch.cuda.Stream(device=dev) for _ in range(max(0, streams - 1))]
        # thread_local: with N > 1 the RCCL watchdog thread polls events while we capture; only THIS thread's calls
        # belong to the capture
        with torch.cuda.graph(g_, capture_error_mode="thread_local"):
            main = torch.cuda.current_stream()
            for st in side:
                st.wait_stream(main)                     # fork
            for i in range(nodes):
                k = i % streams
                if k == 0:
                    step(i, fl)
                else:
                    with torch.cuda.stream(side[k - 1]):
                        step(i, fl)
            for st in side:
                main.wait_stream(st)                     # join
        return g_

    graphs = []          # [(nodes, graph)], longest first
    gn = nb
    if args.launch == "graph":
        for i in range(nb):
            step(i)            # warm (allocates the status word etc. outside capture)
        torch.cuda.synchronize(dev)
        gn = max(1, min(args.graph_nodes, args.steps))
        graphs.append((gn, capture(gn)))
        # exact-size graphs for what is left of K and of W after the long replays: no step of the timed region is
        # launched from Python (a direct call is host-bound at ~15 us), whatever K the caller asks for
        for rem in sorted({args.steps % gn, args.warmup % gn} - {0, gn}, reverse=True):
            graphs.append((rem, capture(rem)))

    def run_steps(k):
        done = 0
        for nodes, g_ in graphs:
            while k - done >= nodes and (nodes == gn or k - done == nodes):
                g_.replay()
                done += nodes
        while done < k:
            step(done)
            done += 1

    # clock / cache pre-warm (not part of the W warmup steps or the K timed steps): ~0.2 s of the same launches,
    # so that a GPU coming out of idle has reached its sustained clock before the contractually timed region
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.2:
        run_steps(max(gn, nb))
        torch.cuda.synchronize(dev)
    run_steps(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    ops.check_status(dev)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Dominant-kernel duration for the roofline object: the SAME launches, strictly sequential on the launch
    # stream (torch's current stream), bracketed by HIP events per group of back-to-back launches (one graph
    # replay or nb direct launches).  The quotient includes the inter-kernel gaps, so it is an upper bound of
    # the kernel's own duration; it is what `rocprofv3 --kernel-trace --stats` reports for the sequential
    # kernel (`siegel_dist_kernel<..., false, false>`).  When the timed region overlaps steps on several
    # streams its kernels live longer individually; that is why the roofline is taken on the sequential pass.
    per_group = gn if graphs else nb
    seq_graph = None
    if graphs and args.streams > 1:
        seq_graph = capture(per_group, streams=1, fl=0)

    def run_group():
        if seq_graph is not None:
            seq_graph.replay()
        else:
            run_steps(per_group)

    for _ in range(2):
        run_group()
    torch.cuda.synchronize(dev)
    groups = max(2, min(16, args.steps // per_group))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    for gidx in range(groups):
        ev[gidx][0].record()
        run_group()
        ev[gidx][1].record()
    torch.cuda.synchronize(dev)
    kernel_ms = sorted(a.elapsed_time(b) / per_group for a, b in ev)
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    kernel_med_ms = kernel_ms[len(kernel_ms) // 2]

    checksum = float(out.sum().item())
    assert checksum == checksum and checksum > 0, "bench produced non-finite distances"

    if rank == 0:
        pairs_total = batch * world * args.steps
        value = pairs_total / elapsed
        bpp = algorithmic_bytes_per_pair(n, model)
        achieved = bpp * batch / (kernel_avg_ms * 1e-3) / 1e9
        traffic = None
        valu_per_wave = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec_pmc = json.load(open(pmc)).get(args.workload, {})
                traffic = rec_pmc.get("hbm_bytes_per_launch")
                c = rec_pmc.get("counters_avg_per_launch", {})
                if c.get("SQ_WAVES"):
                    valu_per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            except Exception:  # noqa: BLE001
                traffic = None
        rec = {
            "metric": "pairwise Siegel distances/sec (upper, riem, n=4)" if args.workload == "upper-riem-n4-b65536"
                      else (f"pairwise SPD affine-invariant distances/sec (n={n})" if model == "spd"
                            else f"pairwise Siegel distances/sec ({model}, {metric}, n={n})"),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "model": None, "manifold": model, "dist_metric": metric,
                       "dims": n, "nodes": nodes, "pairs_per_gpu_per_step": batch,
                       "global_pairs_per_step": batch * world, "table": args.table, "launch": args.launch, "streams": args.streams,
                       "parallelism": f"pairs sharded rank::{world}, table replicated, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "spd16_coop_kernel" if model == "spd" else "siegel_dist_kernel", "kernel_avg_us": kernel_avg_ms * 1e3,
                         "kernel_median_us": kernel_med_ms * 1e3,
                         "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": batch,
                         "pairs_per_s_kernel_only": batch / (kernel_avg_ms * 1e-3),
                         "mode": "sequential launches of the same steps on one stream (rocprof-comparable)"},
            # whole-job throughput of the timed region expressed against the same roof (steps overlap on
            # --streams streams, so this exceeds roofline.frac, which is a per-kernel figure)
            "throughput_frac_of_hbm_roof": value * bpp / (HBM_PEAK_GBS * 1e9),
        }
        # SURVEY 8d's "honest second roofs": the measured copy bandwidth of the chip table, and the fp64 VALU issue
        # slots the whole job occupies (VALU instructions per wave from the committed PMC pass; one wave = 64 pairs;
        # a wave instruction occupies its SIMD for 4 cycles; 1024 SIMDs at the 2.4 GHz peak clock)
        rec["roofline"]["frac_of_measured_copy_bw"] = achieved / MEASURED_COPY_GBS
        if valu_per_wave:
            rec["valu_issue_fraction"] = (value / world) / 64.0 * valu_per_wave * 4.0 / (1024 * 2.4e9)
        del rec["config"]["model"]
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(model, metric, n, nodes, batch, args.seed)
        else:
            rec["cpu_baseline"] = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)      # native stdio buffers go where fd 1 points now (stderr)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(rec) + "\n").encode())


if __name__ == "__main__":
    main()
