#!/usr/bin/env python3
"""bench.py -- pairwise Siegel distances/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (fused Model.forward: gather + distance + metric + scale) over one
batch of synthetic pairs, inputs already resident in HBM.  Default workload: upper / riem / n=4, batch
65 536 pairs per GPU, table of 5 041 nodes (the configuration the headline target is quoted on).

How the K steps reach the GPU (`launch_mode` in the record):
  fused (default, Siegel dims <= 8)  the K batches are handed to `Model.forward_batches` as one list (the
        loop of Runner.evaluate, runner.py:124-135): ONE C call (`sympa_model_forward_batches`), which
        evaluates up to 32 consecutive steps per kernel launch (`siegel_dist_multi_kernel`).  The headline
        pairs/s is therefore that of launches of up to 32 x 65 536 pairs; `roofline` describes that kernel,
        `roofline_default_kernel` the one-launch-per-step kernel a single `Model.forward` call runs.
  graph   one kernel launch per step (C-ABI `sympa_model_forward`), replayed from hipGraphs
  direct  one Python -> C-ABI call and one launch per step

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...` as a CHILD process before it
touches the GPU (train.py:133-136 / README.md:104: the reference is one command too), relays the child's one
JSON line and exits with its return code.

Scaling modes (the pair list shards by triplet, the table is replicated, NO data-path collective):
  weak   (default)  every rank processes its own B-pair shard of a (B x N)-pair global batch
                    (DistributedSampler-style interleave rank::N);
  strong            the global batch is fixed at the workload's B and rank r takes pairs r::N of it
                    (B / N pairs per GPU per step: the reference's semantics, train.py:105-110).

Prints ONE JSON line on rank 0 (DESIGN.md section 7 explains every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (model, metric, dims, nodes, batch)  -- BASELINE.json configs
    "upper-riem-n4-b65536": ("upper", "riem", 4, 5041, 65536),     # headline
    "tree-upper-riem-n4-b8192": ("upper", "riem", 4, 1093, 8192),  # configs[1]
    "grid-upper-riem-n2-b512": ("upper", "riem", 2, 125, 512),     # configs[0]
    "margulis-bounded-finf-n4-b65536": ("bounded", "finf", 4, 5041, 65536),  # configs[2]
    "cartesian-upper-riem-n8-b262144": ("upper", "riem", 8, 45500, 262144),  # configs[3]
    "custom-spd-n16-b1048576": ("spd", "riem", 16, 100000, 1048576),         # configs[4] (parity unpinned: geoopt absent)
}

GRAPH_OF = {   # --pairs graph: the graph whose (i < j, d) triplets the workload trains / evaluates on (BASELINE.json configs)
    "grid-upper-riem-n2-b512": "grid3d-125", "tree-upper-riem-n4-b8192": "tree-b3-h6",
    "margulis-bounded-finf-n4-b65536": "margulis-71", "upper-riem-n4-b65536": "margulis-71",
}

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MEASURED_COPY_GBS = 6290.0   # same guide, chip table: measured copy bandwidth
MODEL_ID = {"upper": 0, "bounded": 1}


def algorithmic_bytes_per_pair(n, model="upper"):
    """2 int64 indices + two fp64 points ([2,n,n] Siegel, [n,n] spd) + one fp64 output, no reuse credit."""
    planes = 1 if model == "spd" else 2
    return 2 * 8 + 2 * (planes * n * n * 8) + 8


def kernel_name(model, n, low_lds, packed=False):
    """Name of the kernel instantiation a forward launch runs (as `rocprofv3 --kernel-trace` prints it)."""
    if model == "spd":
        return f"spd16_coop_kernel<{n}, {'true' if packed else 'false'}>" if n >= 6 else "spd_dist_kernel"
    if n > 8:
        return f"siegel_coop_kernel (n={n})"
    if model == "upper" and n in (7, 8):         # the persistent dense forward (csrc/siegel_packed_kernel.hpp)
        return f"dense_forward_kernel<{n}>"
    low = bool(low_lds) and n in (2, 4)          # DmaTile<N>::ENABLED (csrc/siegel_gather.hpp)
    return f"siegel_dist_kernel<{n}, {MODEL_ID[model]}, {'true' if low else 'false'}>"


def packed_kernel_name(model, n):
    """The kernel sympa_model_forward_packed / _batches_packed launch (csrc/siegel_packed.hip, siegel_packed_kernel.hpp)."""
    return f"packed_forward_kernel<{n}, {MODEL_ID[model]}>"


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_forward_fn(model):
    """The oracle's Model.forward for `model` (test infrastructure: the checker and the CPU baseline, never the product)."""
    import torch
    from oracle import siegel_oracle as so
    if model == "spd":
        one = torch.ones(1, dtype=torch.float64)
        return lambda tab, pr, _model, _metric: so.spd_model_forward(tab, pr, one, 1.0)
    return so.model_forward


def parity_record(got, want, pairs, model):
    """The timed region's own output of batch 0 against the oracle on the same table and pairs (north_star: 1e-4 rel)."""
    import torch
    got, want = got.detach().cpu().to(torch.float64), want.detach().cpu().to(torch.float64)
    finite = bool(torch.isfinite(got).all())
    rel = float(((got - want).abs() / want.abs().clamp_min(1e-9)).max()) if finite else float("inf")
    return {"pairs": int(pairs), "max_rel_err": rel, "tol": 1e-4, "abs_floor": 1e-9, "ok": bool(finite and rel <= 1e-4),
            "against": "oracle/siegel_oracle.py " + ("spd_model_forward (parity UNPINNED: geoopt absent from the reference tree)"
                                                     if model == "spd" else "model_forward (pinned by the imported reference's "
                                                     "goldens, tests/test_oracle_golden.py)") +
                       " on batch 0 of the timed region: same table, same pairs, output taken from the timed launches"}


def cpu_baseline(model, metric, n, nodes, batch, seed, budget_s=12.0, pairs=None, table=None):
    """Oracle (op-for-op torch-CPU fp64 restatement of the reference) timed on the host cores.  Returns (record, the oracle's
    distances of the first `batch` pairs -- kept for the parity object --, that number of pairs)."""
    import torch
    from sympa_amd import data
    oracle_forward = oracle_forward_fn(model)
    if model == "spd":
        batch = min(batch, 65536)
    if table is None:
        table = data.spd_table(nodes, n, seed=seed) if model == "spd" else data.trained_like_table(nodes, n, model=model, seed=seed)
    if pairs is None:
        pairs = data.sample_pairs(nodes, batch, 0, seed)
    else:
        batch = min(batch, pairs.shape[0])
        pairs = pairs[:batch]
    # batched LAPACK eigh on tiny matrices does not scale with threads and collapses
    # when oversubscribed: pick the fastest of a few thread counts on a small probe, report that one
    ncpu = os.cpu_count() or 1
    probe = pairs[: min(batch, 8192)]
    best = (None, 0.0)
    by_threads = {}
    with torch.no_grad():
        # (every logical CPU of a 256-thread host collapses batched LAPACK to ~900 pairs/s and costs 3 s of this leg: capped at 64)
        for threads in sorted({1, min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
            torch.set_num_threads(threads)
            oracle_forward(table, probe[:1024], model, metric)     # warm
            t0 = time.perf_counter()
            oracle_forward(table, probe, model, metric)
            rate = len(probe) / (time.perf_counter() - t0)
            by_threads[threads] = rate
            if rate > best[1]:
                best = (threads, rate)
    torch.set_num_threads(best[0])
    with torch.no_grad():
        done, t0 = 0, time.perf_counter()
        iters = 0
        while True:
            want0 = oracle_forward(table, pairs, model, metric)
            done += batch
            iters += 1
            el = time.perf_counter() - t0
            if el > budget_s or iters >= 50:
                break
    return {"value": done / el, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": ncpu,
            # the 1-thread figure beside the best-of (batched eigh barely scales); probe of len(probe) pairs
            "value_1_thread": by_threads.get(1), "probe_pairs_per_s_by_threads": by_threads,
            "sample": f"{iters} x batch {batch} of the same workload, oracle/siegel_oracle.py "
                      f"{'spd_model_forward' if model == 'spd' else 'model_forward'}, "
                      f"{el:.1f} s"}, want0, batch


LIVE_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_WAVES"))


def live_counters(argv_base, kernel_substring, timeout_s=60):
    """PMC counters of the timed kernel, measured NOW: child runs of this file under `rocprofv3 --pmc <group>` -- FETCH_SIZE,
    WRITE_SIZE, then {SQ_INSTS_VALU, SQ_WAVES}: one counter group per pass, counters only, the program directly after
    `--`, as MI355X_MICROARCH.md prescribes -- every sample normalised by the work-items of its dispatch.  Returns
    {counter: value per work-item, counter + "/dispatch": value per dispatch} for the passes that worked (children, never an
    exec; {} when rocprofv3 is absent).  The per-dispatch figures are what a PERSISTENT kernel needs (the packed forward of dims
    5..8: its grid is the resident waves, not the pairs; the child's launches have the parent's size)."""
    import csv
    import glob
    import shutil
    import tempfile
    tool = shutil.which("rocprofv3")
    per_item = {}
    if tool is None:
        return per_item
    env = dict(os.environ)
    env["SYMPA_BENCH_PMC_CHILD"] = "1"
    env.setdefault("TMPDIR", "/tmp")
    for group in LIVE_PASSES:
        out = tempfile.mkdtemp(prefix="sympa_pmc_")
        try:
            cmd = [tool, "--pmc", *group, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__)] + argv_base
            proc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, timeout=timeout_s,
                                  cwd=tempfile.gettempdir())
            if proc.returncode != 0:
                continue
            total = {c: 0.0 for c in group}
            items = {c: 0.0 for c in group}
            calls = {c: 0 for c in group}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        c = row.get("Counter_Name")
                        if c in total and kernel_substring in row.get("Kernel_Name", ""):
                            total[c] += float(row["Counter_Value"])
                            items[c] += float(row["Grid_Size"])
                            calls[c] += 1
            for c in group:
                if items[c] > 0:
                    per_item[c] = total[c] / items[c]
                    per_item[c + "/dispatch"] = total[c] / calls[c]
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
            continue
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return per_item


def static_flops_per_valu(kernel_key):
    """flops per VALU instruction of a kernel's ISA (FMA = 2; mul / add / min / max / rcp / rsq / sqrt / cvt-free fp64 ops = 1;
    moves, integer and 32-bit work = 0), counted by tools/asm_stats.py::kernel_flop_mix over the assembly the BUILD saved
    (sympa_amd/csrc/asm_stats.json, written by __graft_entry__.build_hip).  None when the file or the kernel is missing."""
    path = os.path.join(ROOT, "sympa_amd", "csrc", "asm_stats.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    for unit in table.values():
        for name, st in unit.items():
            if kernel_key in name and st.get("valu"):
                return {"flops": st["flops"], "valu": st["valu"], "ratio": st["flops"] / st["valu"], "kernel_symbol": name}
    return None


def self_launch(nproc, argv, worker=None):
    """`bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child `torch.distributed.run` (never an
    exec: this is called before anything touches the GPU, and the parent never does), relay the ONE JSON line rank 0
    printed and return the child's exit code.  `worker` (default: this file; SYMPA_BENCH_WORKER overrides it for the CPU
    test, which runs a gloo stub) is the script every rank executes."""
    worker = worker or os.environ.get("SYMPA_BENCH_WORKER") or os.path.abspath(__file__)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC: this pool's host driver supports no other kind (the environment exports HSA_ENABLE_IPC_MODE_LEGACY=0 here and
    # on the GPU boxes, and without it RCCL's cross-process handles fail with `hipIpcGetMemHandle: invalid argument`);
    # setdefault only covers a launch from a scrubbed environment -- a value the caller set is never overridden
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker] + list(argv)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    rc = proc.returncode
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited 0 but printed no JSON record\n")
        rc = 1
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--workload", default="upper-riem-n4-b65536", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch pairs per GPU per step; strong: --batch pairs per step in total, "
                         "rank r takes pairs r::N (train.py:105-110)")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's pairs per step")
    ap.add_argument("--table", default="trained", choices=["trained", "init"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not start the two rocprofv3 --pmc child runs that measure roofline.traffic live (N = 1 only); the "
                         "record then carries the committed counter pass of profiles/pmc_latest.json")
    ap.add_argument("--no-secondary", action="store_true",
                    help="do not append the secondary rows (tools/bench_rows.py: configs[0..4] forward in list and single-call form, "
                         "training steps of the headline / configs[3] / configs[4]) to the record; they are measured by default "
                         "when the headline workload runs on one GPU")
    ap.add_argument("--secondary-budget", type=float, default=150.0, help="seconds the secondary rows may take")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is repeated this many times (each repetition: barrier + synchronize, EXACTLY "
                         "K steps, synchronize; the same warm-up in front of the first); ms_per_step / value are those of the MEDIAN "
                         "repetition (max over ranks per repetition), ms_per_step_min / _max / _all are in the record")
    ap.add_argument("--distinct-batches", type=int, default=16)
    ap.add_argument("--graph-nodes", type=int, default=128,
                    help="kernel launches (= steps) captured per hipGraph")
    ap.add_argument("--streams", type=int, default=0,
                    help="graph launch only: the steps of the timed region are captured on this many parallel HIP "
                         "streams, so that independent steps (different batches, different outputs) overlap on the "
                         "GPU: the next steps gather while the previous ones compute (minimum-LDS kernel form, three "
                         "blocks per CU); 1 = strictly sequential launches; 0 = 4 (8 when a step is < 1 wave per SIMD)")
    ap.add_argument("--launch", default="fused", choices=["fused", "graph", "direct"],
                    help="fused (default; Siegel models with dims <= 8): the K steps go through ONE C call "
                         "(sympa_model_forward_batches, SYMPA_FLAG_FUSE) that evaluates up to --steps-per-launch "
                         "consecutive steps per kernel launch -- a 65 536-pair batch is exactly one wave per SIMD, so the "
                         "steps only fill the machine when several are in flight; graph: one kernel launch per step, "
                         "replayed from captured hipGraphs over --streams parallel streams; direct: one "
                         "Python->C-ABI call and one launch per step")
    ap.add_argument("--steps-per-launch", type=int, default=32, help="fused launch only (<= 32)")
    ap.add_argument("--pairs", default="sampled", choices=["sampled", "graph", "graph-shuffled"],
                    help="sampled: keyed-RNG pairs (i, j != i) of the workload's table; graph / graph-shuffled (configs[0..2] "
                         "only): the workload's real (i < j, d_graph) triplets (preprocess.py:101-126), step j takes the "
                         "j-th batch of them in lexicographic order (the evaluation split, runner.py:124-135) or in "
                         "DistributedSampler order (the training split, train.py:105-110)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    # stdout carries exactly ONE line, the JSON record: everything else that writes to fd 1 -- RCCL prints its version
    # banner and its warnings there from native code -- is sent to stderr for the lifetime of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    from sympa_amd import _lib, data, ops
    from sympa_amd.model import Model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    _lib.load()
    # Test switches (tests/test_multirank_gpu.py: the N > 1 code path with the real kernels on a ONE-GPU box):
    # SYMPA_BENCH_SHARE_GPU=1 puts every rank on cuda:(LOCAL_RANK mod visible devices); SYMPA_BENCH_BACKEND=gloo replaces RCCL
    # (which cannot put two ranks on one device) for the barriers and reductions -- on device tensors all the same;
    # SYMPA_BENCH_DUMP=<dir> saves every rank's batches and outputs.  The driver's runs set none of them.
    device_index = local_rank % torch.cuda.device_count() if os.environ.get("SYMPA_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    backend = os.environ.get("SYMPA_BENCH_BACKEND", "nccl")
    # SYMPA_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, barriers, max-over-ranks) on one GPU
    use_dist = world > 1 or bool(os.environ.get("SYMPA_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

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
    if args.launch == "fused" and (model == "spd" or n > 8):
        args.launch = "graph"          # no fused kernel for the sixteen-lanes-per-pair models
    if args.launch in ("direct", "fused"):
        args.streams = 1
    spl = max(1, min(args.steps_per_launch, ops.MAX_FUSED_BATCHES))
    nb = max(1, min(args.distinct_batches, args.steps))

    # the mirrored reference API: a Model whose table is the synthetic one; the fused timed region hands the K batches to
    # Model.forward_batches as ONE list (plan built on first use, cached on the model)
    class _A:       # the args object Model reads (model.py:8-14)
        manifold, dims, num_points = model, n, nodes
        scale_coef, scale_init, train_scale = 1.0, 1.0, False
    _A.metric = metric
    net = Model(_A)
    with torch.no_grad():
        net.embeddings.embeds.data = table_cpu.clone()
    net = net.to(dev)
    table = net.embeddings.embeds.data      # the single-step kernels below read the same table
    scale = net.scale.data

    spd_pack = net.packed_table() if model == "spd" else None

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    class Region:
        """The K steps of one scaling mode: this rank's shards of `nb` distinct global batches, their outputs, and the launch
        plan (fused list / hipGraphs / direct calls) that runs k steps with no per-step host work."""

        def __init__(self, scaling):
            self.scaling = scaling
            # pairs of one GLOBAL step and of this rank's shard of it
            self.global_pairs = batch * world if scaling == "weak" else batch
            gp = self.global_pairs
            self.batches = []
            if args.pairs == "sampled":
                for j in range(nb):     # global batch j has `gp` pairs; this rank takes the interleave rank::world of it
                    glob = data.sample_pairs(nodes, gp, j, args.seed)
                    self.batches.append(glob[rank::world].contiguous().to(dev))
            else:
                if args.workload not in GRAPH_OF:
                    raise SystemExit(f"--pairs {args.pairs}: no graph for workload {args.workload} (configs[3], [4] sample their "
                                     "pairs, SURVEY 8d)")
                trip_all, id2node = data.graph_triplets(data.named_graph(GRAPH_OF[args.workload]))
                assert len(id2node) == nodes
                if args.pairs == "graph-shuffled":    # the order DistributedSampler(world 1, epoch 0) feeds the training loop
                    order = torch.tensor(data.distributed_sampler_indices(trip_all.shape[0], 1, 0, epoch=0, seed=args.seed))
                    trip_all = trip_all[order]
                total = trip_all.shape[0]
                for j in range(nb):                   # global batch j = triplets [j G, (j+1) G), wrapping around the split
                    rows = (torch.arange(gp) + j * gp) % total
                    self.batches.append(trip_all[rows][rank::world].contiguous().to(dev))
            self.my_pairs = self.batches[0].shape[0]
            self.outs = [torch.empty(self.my_pairs, dtype=torch.float64, device=dev) for _ in range(nb)]
            self.streams = args.streams
            if self.streams <= 0:
                # a launch of < 1 wave per SIMD (65 536 pairs) leaves SIMDs idle: more launches in flight
                self.streams = 4 if self.my_pairs >= 65536 else 8
            self.flags = ops.FLAG_LOW_LDS if (self.streams > 1 and not os.environ.get('SYMPA_BENCH_FULL_LDS')) else 0
            self.flags |= int(os.environ.get('SYMPA_BENCH_FLAGS', '0'), 0)
            # ---- launch plan: K steps = K kernel launches, either direct or replayed from hipGraphs whose nodes are
            # the launches themselves (a step is still exactly one kernel over one batch): a long graph (--graph-nodes
            # launches) plus exact-size graphs for what is left of K and of W, so no step is launched from Python.
            self.graphs = []          # [(nodes, graph)], longest first
            self.gn = nb
            if spd_pack is not None:
                spd_pack.ensure(net.embeddings.embeds)
            for i in range(nb):
                self.step(i)          # warm (allocates the status word etc. outside any capture)
            torch.cuda.synchronize(dev)
            if args.launch == "graph":
                self.gn = max(1, min(args.graph_nodes, args.steps))
                self.graphs.append((self.gn, self.capture(self.gn)))
                for rem in sorted({args.steps % self.gn, args.warmup % self.gn} - {0, self.gn}, reverse=True):
                    self.graphs.append((rem, self.capture(rem)))
            self.lists, self.plans = {}, {}

        def step(self, i, fl=None, dst=None):
            o = (self.outs if dst is None else dst)[i % nb]
            if model == "spd":
                if spd_pack is not None and fl is None:      # Model's no-grad path: the packed table (made / validated once per
                    #                                          K-step call by run_steps, like forward_batches' plan: _SpdBatches.run)
                    ops.spd_model_forward_packed(spd_pack, self.batches[i % nb], scale, 1.0, out=o)
                else:                                        # fl given: the reference pass of the dense kernel
                    ops.spd_model_forward(table, self.batches[i % nb], scale, 1.0, out=o)
                return
            ops.model_forward(table, self.batches[i % nb], model, metric, None, scale, 1.0, out=o,
                              flags=self.flags if fl is None else fl)

        def capture(self, nodes_, streams=None, fl=None, dst=None):
            streams = self.streams if streams is None else streams
            g_ = torch.cuda.CUDAGraph()
            side = [torch.cuda.Stream(device=dev) for _ in range(max(0, streams - 1))]
            # thread_local: with N > 1 the RCCL watchdog thread polls events while we capture; only THIS thread's calls
            # belong to the capture
            with torch.cuda.graph(g_, capture_error_mode="thread_local"):
                main_s = torch.cuda.current_stream()
                for st in side:
                    st.wait_stream(main_s)                     # fork
                for i in range(nodes_):
                    k = i % streams
                    if k == 0:
                        self.step(i, fl, dst)
                    else:
                        with torch.cuda.stream(side[k - 1]):
                            self.step(i, fl, dst)
                for st in side:
                    main_s.wait_stream(st)                     # join
            return g_

        def run_fused(self, k):
            """K steps = ceil(K / spl) launches of up to spl consecutive steps each (step i reads batches[i % nb], writes
            outs[i % nb]).  spl = 32: Model.forward_batches(list of K batches), one C call; another spl (A/B only):
            ops.BatchedForward ranges."""
            if k not in self.lists:
                self.lists[k] = ([self.batches[i % nb] for i in range(k)], [self.outs[i % nb] for i in range(k)])
            bl, ol = self.lists[k]
            if spl == ops.MAX_FUSED_BATCHES:
                # the mirrored API: the list is validated and its pointer arrays built ONCE (Model.prepare_batches, what a
                # caller with a fixed evaluation split does -- Model.evaluate keeps its plan the same way); a call is
                # then one C call.  (Handing the raw lists to forward_batches every time costs ~5 us more per call:
                # the identity key over 2 K tensors.)
                if k not in self.plans:
                    self.plans[k] = net.prepare_batches(bl, ol)
                net.forward_batches(self.plans[k])
                return
            if k not in self.plans:
                self.plans[k] = ops.BatchedForward(table, bl, ol, model, metric, None, scale, 1.0, flags=ops.FLAG_FUSE)
            p = self.plans[k]
            p.set_streams(None)
            for i0 in range(0, k, spl):
                p.run(i0, min(spl, k - i0))

        def run_steps(self, k):
            if args.launch == "fused":
                if k > 0:
                    self.run_fused(k)
                return
            done = 0
            if spd_pack is not None and k > 0:
                spd_pack.ensure(net.embeddings.embeds)      # one validity check (device-side digest) per K-step call
            for nodes_, g_ in self.graphs:
                while k - done >= nodes_ and (nodes_ == self.gn or k - done == nodes_):
                    g_.replay()
                    done += nodes_
            while done < k:
                self.step(done)
                done += 1

        def time(self, repeats):
            """Pre-warm, the W warm-up steps, one untimed pass of the exact launch plan, then `repeats` repetitions of the
            contractually timed region: barrier + synchronize, EXACTLY K steps, synchronize.  Returns this rank's elapsed
            seconds per repetition."""
            # clock / cache pre-warm (not part of the W warmup steps or the K timed steps): ~0.2 s of the same launches,
            # so that a GPU coming out of idle has reached its sustained clock before the contractually timed region
            prewarm_s = float(os.environ.get("SYMPA_BENCH_PREWARM_S", "0.2"))
            pre_graph = None
            if args.launch == "graph" and self.gn < args.graph_nodes and os.environ.get("SYMPA_BENCH_PREWARM_LONG"):
                pre_graph = self.capture(args.graph_nodes)
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < prewarm_s:
                if pre_graph is not None:
                    pre_graph.replay()
                elif args.launch == "fused":
                    self.run_steps(args.steps if args.steps <= 4 * spl else 4 * spl)     # launches of the timed region's shape
                else:
                    self.run_steps(max(self.gn, nb))
                torch.cuda.synchronize(dev)
            self.run_steps(args.warmup)          # the W contractual warmup steps
            if not os.environ.get("SYMPA_BENCH_NO_HOT_REPLAY"):
                self.run_steps(args.steps)       # one more untimed pass of the exact launch plan of the timed region (hot graphs)
            times = []
            for _ in range(max(1, repeats)):
                sync_all()
                t0 = time.perf_counter()
                self.run_steps(args.steps)
                torch.cuda.synchronize(dev)
                times.append(time.perf_counter() - t0)      # this rank's K steps; the MAX over ranks is taken by the caller
            sync_all()                  # closing barrier + synchronize (N > 1: a collective of ~30 us -- every clock is read
            #                             before it, the slowest rank still sets the reported time)
            return times

    def over_ranks(times):
        """Per repetition the MAX over ranks (the reported time) and the MIN (the fastest rank), then the median repetition."""
        t = torch.tensor(times, dtype=torch.float64, device=dev)
        hi, lo = t.clone(), t.clone()
        if use_dist:
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        hi, lo = hi.cpu().tolist(), lo.cpu().tolist()
        order = sorted(range(len(hi)), key=lambda i: hi[i])
        med = order[(len(order) - 1) // 2]          # an actual repetition (the lower median for an even count)
        return {"elapsed": hi[med], "min": min(hi), "max": max(hi), "all": hi, "rank_min_of_median": lo[med],
                "rank_max_of_median": hi[med], "first": hi[0]}

    # ---- the shader clock the chip HOLDS over a stream-ordered region (C-ABI sympa_clock_stamp: s_memtime against the constant
    # 100 MHz s_memrealtime, per XCD): the fp64-issue roof is cycles, and the chip lowers its clock under sustained load
    lib_ = _lib.load()

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import clock_util

    def clock_stamp():
        return clock_util.stamp(lib_, dev, torch)

    clock_between = clock_util.between

    primary = Region(args.scaling)
    batches, outs, my_pairs, global_pairs = primary.batches, primary.outs, primary.my_pairs, primary.global_pairs
    step, capture, run_steps, flags = primary.step, primary.capture, primary.run_steps, primary.flags
    fused = primary if args.launch == "fused" else None
    args.streams = primary.streams

    t_primary = over_ranks(primary.time(args.repeats))
    elapsed = t_primary["elapsed"]
    timed_out0 = outs[0].clone()            # what the TIMED launches wrote for batch 0: the parity object checks this copy
    if os.environ.get("SYMPA_BENCH_DUMP"):
        os.makedirs(os.environ["SYMPA_BENCH_DUMP"], exist_ok=True)
        torch.save({"rank": rank, "world": world, "batches": [t.cpu() for t in batches[:min(nb, args.steps)]],
                    "outs": [t.cpu() for t in outs[:min(nb, args.steps)]], "global_pairs": global_pairs},
                   os.path.join(os.environ["SYMPA_BENCH_DUMP"], f"rank{rank}.pt"))
    # the same K steps once more, untimed by the wall clock, bracketed by HIP events on the launch stream: what the GPU
    # side of such a region takes (recording events INSIDE the wall-timed region costs it ~70 us of host time)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cs0 = clock_stamp()
    ev0.record()
    run_steps(args.steps)
    ev1.record()
    cs1 = clock_stamp()
    torch.cuda.synchronize(dev)
    device_ms = ev0.elapsed_time(ev1)
    clock_region = clock_between(cs0, cs1)       # over one K-step region entered from an idle GPU, like the wall-timed ones
    ops.check_status(dev)
    ranks_seen = 1
    if use_dist:
        t = torch.tensor([device_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        device_ms = float(t[0].item())
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)       # every rank that took part in the timed region counts itself
        ranks_seen = int(ones.item())
    # N > 1: the OTHER scaling mode in the same command -- `--scaling weak` (the default: B pairs per GPU per step) holds
    # ">= 6x at 8 GPUs" by construction, the strong figure (the workload's B pairs per step in total, rank r takes r::N: the
    # reference's semantics, train.py:105-110, per-process batch = batch / n_procs) is the one that shows scaling
    t_other, other = None, None
    if use_dist and not os.environ.get("SYMPA_BENCH_ONE_SCALING"):
        other = Region("strong" if args.scaling == "weak" else "weak")
        t_other = over_ranks(other.time(args.repeats))
        ops.check_status(dev)

    # ---- every output of the timed variant against a strictly sequential pass of the default kernel form
    # (different instantiation when the timed region ran the minimum-LDS form): same arithmetic => bit-identical
    # up to the wave-dependent number of extra (harmless) QL sweeps for n >= 5
    ref_outs = [torch.empty_like(o) for o in outs]
    for i in range(nb):
        step(i, 0, ref_outs)
    torch.cuda.synchronize(dev)
    for i in range(min(nb, args.steps)):
        same = torch.equal(outs[i], ref_outs[i]) if (model != "spd" and n <= 4) else \
            torch.allclose(outs[i], ref_outs[i], rtol=1e-11, atol=1e-13)
        assert same, f"timed-region output {i} differs from the sequential default-kernel pass"
    checksum = float(outs[0].sum().item())
    assert checksum == checksum and checksum > 0, "bench produced non-finite distances"

    # ---- per-kernel duration for the roofline objects: the SAME launches, strictly sequential on the launch
    # stream (torch's current stream), bracketed by HIP events per group of back-to-back launches.  The quotient
    # includes the inter-kernel gaps (upper bound of the kernel's own duration); it is what
    # `rocprofv3 --kernel-trace --stats` reports per kernel (the profiler serialises a queue's dispatches).
    per_group = max(1, min(args.graph_nodes, max(args.steps, 64)))

    def timed_groups(run_group, launches_per_group):
        for _ in range(2):
            run_group()
        torch.cuda.synchronize(dev)
        groups = 8
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
        for a, b in ev:
            a.record()
            run_group()
            b.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) / launches_per_group for a, b in ev)
        return sum(ms) / len(ms), ms[len(ms) // 2]

    def kernel_time(fl):
        """Average duration of ONE single-step launch with gather flags `fl`, launches strictly sequential."""
        if args.launch == "direct":
            def run_group():
                for i in range(per_group):
                    step(i, fl)
        else:
            run_group = capture(per_group, streams=1, fl=fl).replay
        return timed_groups(run_group, per_group)

    default_kernel = kernel_name(model, n, 0)
    pack_us = None
    if fused is not None:
        # the timed region's kernel: launches of spl steps (the last one of what is left), strictly sequential on the
        # launch stream -- its average duration IS the event-bracketed repetition of the timed region over its launches
        n_launches = (args.steps + spl - 1) // spl
        packed = spl == ops.MAX_FUSED_BATCHES and net.packed_table() is not None and my_pairs * args.steps >= 4096
        # dims 5..8: Model.forward_batches runs the list over the PACKED table (one pack per table version, ops.PackedTable)
        timed_kernel = (packed_kernel_name(model, n) if packed else f"siegel_dist_multi_kernel<{n}, {MODEL_ID[model]}>")
        # >= 8 event-bracketed groups (average and median), not one sample.  A group is enough back-to-back repetitions of
        # the timed region's launches (>= 64 launches) that the host-side cost of a call hides behind the kernels of the
        # previous one: the quotient is then the kernel's own duration, the figure rocprofv3 --kernel-trace reports
        reps = max(1, -(-64 // n_launches))

        def fused_group():
            for _ in range(reps):
                run_steps(args.steps)
        k_timed = timed_groups(fused_group, reps * n_launches)
        cs0 = clock_stamp()
        for _ in range(4):
            fused_group()
        cs1 = clock_stamp()
        torch.cuda.synchronize(dev)
        clock_sustained = clock_between(cs0, cs1)       # back-to-back launches of the timed kernel: what kernel_avg_us was measured at
        timed_pairs_per_launch = my_pairs * args.steps / n_launches
        k_default = kernel_time(0)
        if packed:          # what a table that changes before every call pays on top: one sympa_table_pack over the table
            pk = net.packed_table()

            def repack():
                pk.invalidate()
                pk.ensure(net.embeddings.embeds)
            pack_us = timed_groups(repack, 1)[1] * 1e3
    else:
        timed_kernel = kernel_name(model, n, flags & ops.FLAG_LOW_LDS, packed=spd_pack is not None)
        k_timed = kernel_time(None if spd_pack is not None else flags)
        cs0 = clock_stamp()
        run_steps(max(args.steps, 64))
        cs1 = clock_stamp()
        torch.cuda.synchronize(dev)
        clock_sustained = clock_between(cs0, cs1)
        timed_pairs_per_launch = my_pairs
        k_default = k_timed if default_kernel == timed_kernel else kernel_time(0)
        if spd_pack is not None:
            def repack():
                spd_pack.invalidate()
                spd_pack.ensure(net.embeddings.embeds)
            pack_us = timed_groups(repack, 1)[1] * 1e3

    if rank == 0:
        pairs_total = global_pairs * args.steps
        value = pairs_total / elapsed
        bpp = algorithmic_bytes_per_pair(n, model)
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path)).get(args.workload, {})
            except Exception:  # noqa: BLE001
                pmc = {}
        c = pmc.get("counters_avg_per_launch", {})
        committed_valu_per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"] if c.get("SQ_WAVES") else None
        # the committed counter pass of THIS command line (tools/pmc_collect.sh): the fallback when nothing can be measured
        # now (stored per step = per 65 536-pair batch; a fused launch moves that times its steps)
        traffic = pmc.get("hbm_bytes_per_step", pmc.get("hbm_bytes_per_launch")) \
            if my_pairs == WORKLOADS[args.workload][4] else None

        # ... measured now: child runs of this command line under `rocprofv3 --pmc` (N = 1, not when this process is itself
        # a PMC child or runs under a profiler): FETCH_SIZE, WRITE_SIZE, {SQ_INSTS_VALU, SQ_WAVES}
        live, counters, t_live = None, {}, 0.0
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower()
        if (world == 1 and not args.no_live_traffic and not os.environ.get("SYMPA_BENCH_PMC_CHILD") and not profiled
                and " (" not in timed_kernel):          # (the dims 9..16 entry is a description, not a kernel name)
            child = ["--gpus", "1", "--workload", args.workload, "--steps", str(min(args.steps, 128)),
                     "--warmup", str(min(args.warmup, 32)), "--no-cpu-baseline", "--no-live-traffic", "--launch", args.launch,
                     "--pairs", args.pairs, "--table", args.table, "--scaling", args.scaling,
                     "--steps-per-launch", str(args.steps_per_launch), "--streams", str(args.streams)]
            if args.batch:
                child += ["--batch", str(args.batch)]
            t_live = time.perf_counter()
            counters = live_counters(child, timed_kernel.split(" (")[0])
            t_live = time.perf_counter() - t_live
            persistent = timed_kernel.startswith("packed_forward")
            if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                if persistent:      # (the child's launches are the parent's as long as both run whole 32-step groups)
                    live = (2.0 * counters["FETCH_SIZE/dispatch"] + counters["WRITE_SIZE/dispatch"]) * 1024.0
                    if min(args.steps, 128) % spl or args.steps % spl:
                        live = None
                else:
                    live = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0 * timed_pairs_per_launch
        # work-items per pair of the timed kernel's grid: the spd kernel's grid is one work-item per pair (a wave serves its
        # 64 pairs four at a time, sixteen lanes each, over sixteen rounds); the Siegel dims 9..16 kernels launch sixteen
        lanes_per_pair = 16 if (model != "spd" and n > 8) else 1
        if counters.get("SQ_WAVES") and timed_kernel.startswith("packed_forward"):
            # persistent waves, each walking many tiles of 64 pairs: wave-instructions per PAIR from the per-dispatch totals,
            # expressed as "per wave of 64 pairs" like the other kernels
            valu_per_wave = counters["SQ_INSTS_VALU/dispatch"] / timed_pairs_per_launch * 64.0
            waves_per_pair = 1.0 / 64.0
            valu_source = (f"measured in this run: a child run of the same command line under `rocprofv3 --pmc SQ_INSTS_VALU "
                           f"SQ_WAVES` (counters only), samples of {timed_kernel} per dispatch / pairs per launch x 64 (persistent "
                           "waves: VALU instructions per 64 pairs)")
            if min(args.steps, 128) % spl or args.steps % spl:
                valu_per_wave, valu_source = None, None
        elif counters.get("SQ_WAVES"):
            valu_per_wave = counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"]
            waves_per_pair = counters["SQ_WAVES"] * lanes_per_pair        # counters are per work-item
            valu_source = (f"measured in this run: a child run of the same command line under `rocprofv3 --pmc SQ_INSTS_VALU "
                           f"SQ_WAVES` (counters only), samples of {timed_kernel} normalised by the work-items of their dispatches")
        else:
            valu_per_wave = committed_valu_per_wave
            waves_per_pair = lanes_per_pair / 64.0
            valu_source = (f"profiles/pmc_latest.json [{pmc.get('round', '?')}]: a committed counter pass, not measured in this run"
                           if valu_per_wave else None)

        def roof(kname, kt, pairs_per_launch, note):
            avg_ms, med_ms = kt
            ach = bpp * pairs_per_launch / (avg_ms * 1e-3) / 1e9
            if live is not None and kname == timed_kernel:
                tr = live
                src = (f"measured in this run: two child runs of the same command line under `rocprofv3 --pmc FETCH_SIZE` / "
                       f"`--pmc WRITE_SIZE` (separate passes, counters only; all PMC passes {t_live:.0f} s), samples of {kname} normalised by "
                       "the work-items of their dispatches; 2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950 half-count of 16 B/lane "
                       "reads, MI355X_MICROARCH.md HBM section)")
            else:
                tr = traffic * (pairs_per_launch / my_pairs) if traffic else None
                src = (f"profiles/pmc_latest.json [{pmc.get('round', '?')}]: {pmc.get('source', '')}; "
                       "2 x FETCH_SIZE + WRITE_SIZE (gfx950 half-count of 16 B/lane reads, "
                       "MI355X_MICROARCH.md HBM section), per single-step launch x steps per launch; "
                       "a committed counter pass, not measured in this run") if tr else None
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": tr,
                    "traffic_source": src,
                    "kernel": kname, "kernel_avg_us": avg_ms * 1e3, "kernel_median_us": med_ms * 1e3,
                    "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": pairs_per_launch,
                    "pairs_per_s_kernel_only": pairs_per_launch / (avg_ms * 1e-3),
                    "frac_of_measured_copy_bw": ach / MEASURED_COPY_GBS,
                    "mode": note}

        rec = {
            "metric": "pairwise Siegel distances/sec (upper, riem, n=4)" if args.workload == "upper-riem-n4-b65536"
                      else (f"pairwise SPD affine-invariant distances/sec (n={n})" if model == "spd"
                            else f"pairwise Siegel distances/sec ({model}, {metric}, n={n})"),
            "value": value, "unit": "pairs/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup,
            # the MEDIAN of `repetitions` repetitions of the K-step timed region (each: barrier + synchronize, exactly K steps,
            # synchronize; max over ranks per repetition); the spread and the first repetition are beside it
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": t_primary["min"] / args.steps * 1e3, "ms_per_step_max": t_primary["max"] / args.steps * 1e3,
            "ms_per_step_first": t_primary["first"] / args.steps * 1e3,
            "ms_per_step_all": [x / args.steps * 1e3 for x in t_primary["all"]], "repetitions": len(t_primary["all"]),
            # fastest and slowest rank of the median repetition (K steps, milliseconds)
            "elapsed_ranks_ms": {"min": t_primary["rank_min_of_median"] * 1e3, "max": t_primary["rank_max_of_median"] * 1e3},
            # HIP events on the launch stream around an identical repetition of the K steps (max over ranks): what the GPU
            # side of the timed region takes, launch latency of the first graph included
            "ms_per_step_device": device_ms / args.steps,
            # wall-clock-free throughput of the same region (the ~10 us of launch + sync latency of a K = 20 region excluded)
            "value_device": global_pairs * args.steps / (device_ms * 1e-3),
            "launch_mode": args.launch, "steps_per_launch": (min(spl, args.steps) if fused is not None else 1),
            "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "manifold": model, "dist_metric": metric,
                       "dims": n, "nodes": nodes, "pairs_per_gpu_per_step": my_pairs,
                       "global_pairs_per_step": global_pairs, "table": args.table, "pairs": args.pairs,
                       "launch": args.launch,
                       "api": ("Model.forward_batches(Model.prepare_batches(list of K batches))" if (fused is not None and
                                                                                spl == ops.MAX_FUSED_BATCHES)
                               else "ops (C-ABI binding) directly"),
                       "streams": args.streams,
                       "parallelism": f"pairs sharded rank::{world}, table replicated, no collective"},
            # the kernel the timed region ran, its launches strictly sequential (what rocprofv3 --kernel-trace reports)
            "roofline": roof(timed_kernel, k_timed, timed_pairs_per_launch,
                             ("the timed region's kernel: one launch evaluates up to %d consecutive steps (batches); duration = "
                              "HIP-event time of 8 groups of back-to-back repetitions of the timed region (%d launch(es) each, "
                              ">= 64 launches per group) / launches; " % (spl, (args.steps + spl - 1) // spl)
                              if fused is not None else "the timed region's kernel instantiation, one step per launch, "
                              "launches strictly sequential on one stream, HIP events per group; ")
                             + "rocprof-comparable. CONTRACT roof (SURVEY 8d): algorithmic bytes (536 B/pair at n = 4, no "
                             "reuse credit) against the 8 TB/s HBM peak -- the table is cache-resident (`traffic` is what "
                             "actually reaches the fabric), so this fraction can approach or exceed 1; the physical "
                             "limiter is fp64 VALU issue: see valu_issue_fraction and DESIGN.md section 5"),
            # whole-job throughput of the timed region expressed against the same roof
            "throughput_frac_of_hbm_roof": (value / world) * bpp / (HBM_PEAK_GBS * 1e9),
        }
        rec["config"]["steps_per_launch"] = min(spl, args.steps) if fused is not None else 1
        if pack_us is not None:
            rec["packed_table"] = {"pack_us": pack_us, "rows": nodes,
                                   "ms_per_step_if_the_table_changed_before_every_step": rec["ms_per_step"] + pack_us * 1e-3,
                                   "note": "Siegel dims 5..8 / spd: the timed region reads the packed table (per point the upper triangles "
                                           "+ inverted Cholesky factor, spd: the unit LDL^T factor; made once per table version: "
                                           "sympa_table_pack / sympa_spd_table_pack); the pack is NOT in the "
                                           "timed region (the table does not change between the K steps, as in Runner.evaluate, "
                                           "runner.py:124-135) -- its cost is reported here"}
        rec["value_" + args.scaling] = value
        rec["ms_per_step_" + args.scaling] = rec["ms_per_step"]
        if t_other is not None:
            o = other.scaling
            rec["value_" + o] = other.global_pairs * args.steps / t_other["elapsed"]
            rec["ms_per_step_" + o] = t_other["elapsed"] / args.steps * 1e3
            rec["ms_per_step_" + o + "_min"] = t_other["min"] / args.steps * 1e3
            rec["ms_per_step_" + o + "_max"] = t_other["max"] / args.steps * 1e3
            rec["elapsed_ranks_ms_" + o] = {"min": t_other["rank_min_of_median"] * 1e3, "max": t_other["rank_max_of_median"] * 1e3}
            rec["config"]["pairs_per_gpu_per_step_" + o] = other.my_pairs
            rec["config"]["global_pairs_per_step_" + o] = other.global_pairs
        if default_kernel != timed_kernel:
            rec["roofline_default_kernel"] = roof(
                default_kernel, k_default, my_pairs,
                "the kernel ONE Model.forward call runs (one step per launch, both endpoints staged at once, one block "
                "per CU: a 65 536-pair batch is one wave per SIMD), launches strictly sequential on one stream")
        # SURVEY 8d's "honest second roof", the PHYSICAL one: fp64 VALU issue slots.  A wave instruction occupies its SIMD
        # for 4 cycles (16 fp64 lanes per SIMD and clock); 1 024 SIMDs at the 2.4 GHz peak clock = 614.4 G wave-instructions/s.
        # `frac` is the timed kernel's own (its launches strictly sequential: VALU instructions per wave x waves per launch /
        # kernel duration); `frac_whole_job` the same for the wall-clock throughput of the timed region.
        VALU_PEAK = 1024 * 2.4e9 / 4.0
        if valu_per_wave:
            mix = static_flops_per_valu(timed_kernel.split(" (")[0])
            waves_per_launch = timed_pairs_per_launch * waves_per_pair
            k_s = k_timed[0] * 1e-3
            ach = valu_per_wave * waves_per_launch / k_s
            phys = {"bound": "fp64_valu", "achieved": ach / 1e9, "peak": VALU_PEAK / 1e9, "unit": "G wave-instructions/s",
                    "frac": ach / VALU_PEAK,
                    "frac_whole_job": (value / world) * waves_per_pair * valu_per_wave / VALU_PEAK,
                    "kernel": timed_kernel, "kernel_avg_us": k_timed[0] * 1e3,
                    "valu_instructions_per_wave": valu_per_wave, "waves_per_launch": waves_per_launch,
                    "source": valu_source,
                    "note": "issue-slot fraction at the PEAK clock: the chip holds a lower clock under sustained fp64 load "
                            "(MI355X_MICROARCH.md, DVFS), so 1.0 is not reachable; every VALU instruction counts (fp64 "
                            "arithmetic, moves, integer work)"}
            if mix is not None:
                own = valu_per_wave * waves_per_pair * 64.0 * mix["ratio"]      # every lane of the pair's wave(s) counts
                phys["flops"] = {"peak_tflops": 78.6, "own_flops_per_pair": own,
                                 "frac_own": (timed_pairs_per_launch / k_s) * own / 78.6e12,
                                 "static_flops_per_valu_instruction": mix["ratio"],
                                 "static_mix": f"{mix['flops']} flops in {mix['valu']} VALU instructions of {mix['kernel_symbol']} "
                                               "(sympa_amd/csrc/asm_stats.json, written by the build from the saved assembly: "
                                               "FMA = 2, other fp64 arithmetic = 1, everything else 0)",
                                 "reference_flops_per_pair": (121.0 * n ** 3 if model != "spd" else None)}
            if clock_sustained:
                # the same quotient against the issue slots of the clock the chip actually held while the kernel ran back to back
                peak_now = 1024 * clock_sustained["mhz"] * 1e6 / 4.0
                phys["frac_at_measured_clock"] = ach / peak_now
                phys["measured_clock_mhz"] = clock_sustained["mhz"]
                if clock_region:
                    phys["frac_whole_job_at_measured_clock"] = ((value / world) * waves_per_pair * valu_per_wave /
                                                                (1024 * clock_region["mhz"] * 1e6 / 4.0))
            rec["roofline_physical"] = phys
            rec["valu_issue_fraction"] = phys["frac_whole_job"]
            if mix is not None:
                rec["fp64_flops"] = {"peak_tflops": 78.6, "own_flops_per_pair": phys["flops"]["own_flops_per_pair"],
                                     "reference_flops_per_pair": phys["flops"]["reference_flops_per_pair"],
                                     "frac_own": (value / world) * phys["flops"]["own_flops_per_pair"] / 78.6e12,
                                     "source": valu_source}
        rec["clock"] = {"spec_peak_mhz": 2400.0, "sustained": clock_sustained, "timed_region": clock_region,
                        "method": "C-ABI sympa_clock_stamp before and after the region on the launch stream: 2 048 one-wave blocks "
                                  "each record s_memtime (shader cycles), s_memrealtime (constant 100 MHz) and the CU they ran on; clock = "
                                  "d(s_memtime) / d(s_memrealtime) x 100 MHz between the stamps of the SAME CU (MI355X_MICROARCH.md, DVFS "
                                  "give-back item 6), median over the CUs.  sustained: around back-to-back launches of the timed kernel (the region "
                                  "kernel_avg_us comes from); timed_region: around ONE K-step region entered from an idle GPU "
                                  "(includes the launch gap in front of its first kernel)"}
        # ---- parity of the timed region's own output (batch 0) against the oracle, same table, same pairs
        want0 = None
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"], want0, k0 = cpu_baseline(model, metric, n, nodes, my_pairs, args.seed, table=table_cpu,
                                                          pairs=batches[0][:, :2].cpu())
        else:
            rec["cpu_baseline"] = None
            k0 = min(my_pairs, 4096)            # no timed baseline: the checker still runs, on a 4 096-pair sample
            with torch.no_grad():
                want0 = oracle_forward_fn(model)(table_cpu, batches[0][:k0, :2].cpu(), model, metric)
        rec["parity"] = parity_record(timed_out0[:k0], want0, k0, model)
        # ---- secondary rows (tools/bench_rows.py): the other configs' forward in both call forms and the training steps,
        # measured by THIS command, in this JSON line
        if (world == 1 and args.workload == "upper-riem-n4-b65536" and not args.no_secondary and not args.batch
                and not os.environ.get("SYMPA_BENCH_PMC_CHILD") and not profiled):
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_rows
            t_sec = time.perf_counter()
            rec["secondary"] = bench_rows.secondary_rows(dev, args.seed, steps=min(args.steps, 20), budget_s=args.secondary_budget,
                                                         log=lambda s_: sys.stderr.write(s_ + "\n"))
            rec["secondary_seconds"] = round(time.perf_counter() - t_sec, 1)
            rec["secondary_legend"] = bench_rows.LEGEND
            bad_rows = [r for r in rec["secondary"] if "parity" in r and not r["parity"]["ok"]]
            if bad_rows:
                sys.stderr.write("bench.py: secondary rows with a parity failure: " + ", ".join(
                    f"{r['workload']}/{r['kind']}/{r.get('form', '')}" for r in bad_rows) + "\n")
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
        if not rec["parity"]["ok"]:
            sys.stderr.write(f"bench.py: PARITY FAILURE: max rel err {rec['parity']['max_rel_err']:.3e} > 1e-4 against the oracle\n")
            raise SystemExit(3)


if __name__ == "__main__":
    main()
