"""CPU, world_size 2 over gloo: the N>1 path of the forward metric.  The path shards by pair with no
data-path collective, so what has to be right is the index logic: every rank
derives its shard of each global batch from the same keyed RNG, the shards are disjoint, their union
is the single-rank batch, DistributedSampler semantics hold, and the max-over-ranks timing reduction
works.  (The kernels themselves need a GPU; the oracle stands in for the arithmetic here.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import siegel_oracle as so
from sympa_amd import data


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nodes, n, batch = 200, 3, 256
        table = data.trained_like_table(nodes, n, seed=42)           # replicated, identical bytes
        glob = data.sample_pairs(nodes, batch * world, 5, seed=42)   # global batch 5
        mine = glob[rank::world].contiguous()                        # bench.py's sharding
        d = so.model_forward(table, mine, "upper", "riem")
        # gather the shards on rank 0 (test-only collective) and compare with the single-rank result
        outs = [torch.zeros_like(d) for _ in range(world)]
        idxs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(outs, d)
        dist.all_gather(idxs, mine)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                     # bench.py's max-over-ranks
        # strong scaling (bench.py --scaling strong; the reference's semantics, train.py:105-110): the global batch
        # keeps the single-rank size, rank r takes pairs r::world of it
        glob_s = data.sample_pairs(nodes, batch, 5, seed=42)
        mine_s = glob_s[rank::world].contiguous()
        d_s = so.model_forward(table, mine_s, "upper", "riem")
        outs_s = [torch.zeros_like(d_s) for _ in range(world)]
        dist.all_gather(outs_s, d_s)
        # DistributedSampler semantics for a triplet list (train.py:105-110)
        shard = data.distributed_sampler_indices(1001, world, rank, epoch=3, seed=0)
        all_shards = [None] * world
        dist.all_gather_object(all_shards, shard)
        if rank == 0:
            full = so.model_forward(table, glob, "upper", "riem")
            merged = torch.empty_like(full)
            merged_idx = torch.empty_like(glob)
            for r in range(world):
                merged[r::world] = outs[r]
                merged_idx[r::world] = idxs[r]
            results["union_equal"] = bool(torch.equal(merged_idx, glob))
            results["dist_equal"] = bool(torch.equal(merged, full))
            results["max"] = float(t.item())
            full_s = so.model_forward(table, glob_s, "upper", "riem")
            merged_s = torch.empty_like(full_s)
            for r in range(world):
                merged_s[r::world] = outs_s[r]
            results["strong_equal"] = bool(torch.equal(merged_s, full_s)) and mine_s.shape[0] * world == batch
            flat = sorted(i for s in all_shards for i in s)
            per = -(-1001 // world)                    # DistributedSampler pads the list to a multiple of the world size
            results["sampler_cover"] = sorted(set(flat)) == list(range(1001)) and len(flat) == per * world
            results["sampler_sizes"] = [len(s) for s in all_shards]
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_over_gloo():
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert results["union_equal"] and results["dist_equal"] and results["strong_equal"]
    assert results["max"] == 2.0
    assert results["sampler_cover"] and results["sampler_sizes"] == [501, 501]


def test_eight_rank_sharding_over_gloo():
    """World size 8 (round-4 review: rank::8 had never run): the interleave, the strong-scaling shards of 256 / 8 pairs and the
    sampler's padding of 1 001 triplets to 8 x 126."""
    world = 8
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert results["union_equal"] and results["dist_equal"] and results["strong_equal"]
    assert results["max"] == 8.0
    assert results["sampler_cover"] and results["sampler_sizes"] == [126] * 8


def _grad_worker(rank, world, port, results):
    """Data-parallel training semantics over gloo: per-rank gradients of the rank's shard, one flat
    all-reduce (mean), must equal DDP's result = gradient of the mean-of-shards loss."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympa_amd.distributed import allreduce_gradients, shard_triplets
        nodes, n = 40, 2
        table0 = data.trained_like_table(nodes, n, seed=7)
        trip = torch.cat((data.sample_pairs(nodes, 128, 0, seed=7), torch.randint(1, 6, (128, 1))), 1)
        if rank == 0:
            obj = [trip]
        else:
            obj = [None]
        dist.broadcast_object_list(obj, src=0)
        trip = obj[0]
        mine = shard_triplets(trip, rank, world, epoch=1, seed=0)

        def loss_of(table, t):
            d = so.model_forward(table, t, "upper", "riem")
            return so.distortion_loss(t[:, 2].to(torch.float64), d)

        table = torch.nn.Parameter(table0.clone())
        scale = torch.nn.Parameter(torch.ones(1, dtype=torch.float64))
        loss_of(table, mine).backward()          # scale has no grad on purpose (treated as zeros)
        allreduce_gradients([table, scale])
        if rank == 0:
            ref = torch.nn.Parameter(table0.clone())
            total = sum(loss_of(ref, shard_triplets(trip, r, world, epoch=1, seed=0)) for r in range(world)) / world
            total.backward()
            results["grad_equal"] = bool(torch.allclose(table.grad, ref.grad, rtol=1e-12, atol=1e-14))
            results["scale_zero"] = bool(torch.all(scale.grad == 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_ddp_mean():
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_grad_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert results["grad_equal"] and results["scale_zero"]


def _exchange_worker(rank, world, port, results):
    """GradientExchange over gloo: the dense mode (flat persistent buffer, in-place all-reduce) and the touched-row
    mode (all-gather of per-pair rows + indices, scatter-add) both give DDP's mean gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sympa_amd.distributed import GradientExchange, shard_triplets
        nodes, n, b_glob = 300, 2, 64                      # 2 B = 128 < N = 300: the rows message is the smaller one
        table0 = data.trained_like_table(nodes, n, seed=9)
        g = torch.Generator().manual_seed(1)
        trip = torch.cat((data.sample_pairs(nodes, b_glob, 0, seed=9), torch.randint(1, 6, (b_glob, 1), generator=g)), 1)
        trip[5, 0] = trip[6, 1]                            # a node touched twice inside one shard and across shards
        trip[7, 0] = trip[6, 1]
        mine = shard_triplets(trip, rank, world, epoch=2, seed=0)
        b = mine.shape[0]

        def reference_mean_grad():
            ref = torch.nn.Parameter(table0.clone())
            sc = torch.nn.Parameter(torch.full((1,), 1.7, dtype=torch.float64))
            total = 0.0
            for r in range(world):
                t = shard_triplets(trip, r, world, epoch=2, seed=0)
                d = so.model_forward(ref, t, "upper", "riem", None, sc, 1.0)
                total = total + so.distortion_loss(t[:, 2].to(torch.float64), d)
            (total / world).backward()
            return ref.grad, sc.grad

        for mode in ("dense", "rows", "auto"):
            table = torch.nn.Parameter(table0.clone())
            scale = torch.nn.Parameter(torch.full((1,), 1.7, dtype=torch.float64))
            ex = GradientExchange([table, scale], table=table, local_batch=b, mode=mode,
                                  scatter_fn=lambda gt, rows, idx, alpha: gt.index_add_(0, idx, rows.view(-1, *gt.shape[1:]), alpha=alpha))
            assert table.grad.data_ptr() == ex.flat.data_ptr()
            ex.zero_()
            if ex.mode == "dense":
                d = so.model_forward(table, mine, "upper", "riem", None, scale, 1.0)
                so.distortion_loss(mine[:, 2].to(torch.float64), d).backward()      # accumulates into the views
                ex.check_views()
                ex.allreduce()
            else:
                sc = scale.detach().clone().requires_grad_(True)
                z1 = table.detach()[mine[:, 0]].clone().requires_grad_(True)
                z2 = table.detach()[mine[:, 1]].clone().requires_grad_(True)
                dd = so.manifold_dist("upper", z1, z2, "riem") * (sc / 1.0).clamp_min(0.1)
                so.distortion_loss(mine[:, 2].to(torch.float64), dd).backward()
                ex.rows.copy_(torch.cat((z1.grad, z2.grad), 0).reshape(2 * b, -1))
                scale.grad.add_(sc.grad)
                ex.exchange_rows(mine[:, 0], mine[:, 1])
            want_t, want_s = reference_mean_grad()
            ok = torch.allclose(table.grad, want_t, rtol=1e-11, atol=1e-13) and \
                torch.allclose(scale.grad, want_s, rtol=1e-11, atol=1e-13)
            if rank == 0:
                results[f"{mode}_ok"] = bool(ok)
                results[f"{mode}_mode"] = ex.mode
                results[f"{mode}_bytes"] = ex.message_bytes
        if rank == 0:
            results["dense_bound"] = nodes * 16 * n * n
            results["rows_bound"] = 2 * b_glob * 16 * n * n
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange_dense_and_touched_rows():
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert results["dense_ok"] and results["rows_ok"] and results["auto_ok"]
    assert results["auto_mode"] == "rows"                       # 2 B < N
    # per-step message: touched rows (+ their int64 indices + the scalar) stay below 2 B 16 n^2 (1 + 1/(2 n^2)) and
    # below the dense table
    assert results["rows_bytes"] <= results["rows_bound"] * 1.2 + 64
    assert results["rows_bytes"] < results["dense_bytes"]


def _run_bench_self_launch(extra_env, argv):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["SYMPA_BENCH_WORKER"] = os.path.join(root, "tests", "bench_stub_worker.py")
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_bench_gpus_n_launches_its_own_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2 --steps K --warmup W` (the driver's command shape, no torchrun around it) starts the
    ranks itself as a child torch.distributed.run -- before any GPU call: this container has no GPU and the parent
    never needs one -- and relays rank 0's single JSON line.  The rank body is a gloo stub here."""
    import json
    p = _run_bench_self_launch({}, ["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5
    assert rec["max_over_ranks"] == 2.0


def test_bench_gpus_8_launches_eight_ranks():
    """The 8-GPU command of the driver's scaling run, with the gloo stub as rank body: eight ranks, one JSON line."""
    import json
    p = _run_bench_self_launch({}, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and rec["max_over_ranks"] == 8.0


def test_bench_self_launch_propagates_a_failing_rank():
    p = _run_bench_self_launch({"STUB_FAIL_RANK": "1"}, ["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert p.returncode != 0
    assert p.stdout.decode().strip() == "" or "stub" in p.stdout.decode()
