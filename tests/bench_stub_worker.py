"""Stand-in for bench.py's rank body in the CPU test of its self-launch path (tests/test_multirank_cpu.py): every
rank joins a gloo group from the environment torch.distributed.run prepared, the ranks count themselves with an
all-reduce, rank 0 prints noise and then ONE JSON line.  STUB_FAIL_RANK makes that rank exit with code 7."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == args.gpus, (world, args.gpus)
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
ones = torch.ones(1, dtype=torch.int64)
dist.all_reduce(ones)
t = torch.tensor([1.0 + rank])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if rank == 0:
    print("noise that is not the record")
    print(json.dumps({"metric": "stub", "n_gpus": world, "ranks_seen": int(ones.item()), "steps": args.steps,
                      "warmup": args.warmup, "max_over_ranks": float(t.item())}))
