"""Data-parallel training glue.  One process per GPU; pairs are sharded by triplet with
DistributedSampler semantics, the table is replicated, and the only exchange step of the path is the
all-reduce of the gradients after backward.

The reference wraps the model in torch DDP (train.py:59), which all-reduces (mean) the dense table
gradient [N,2,n,n] plus the scalars bucket by bucket on every backward (runner.py:105).  That still works
with this package (the autograd Functions return ordinary dense gradients).  The fused training kernel
(`Model.fused_loss_backward`) writes `.grad` without going through autograd hooks, so this module offers
the equivalent exchange explicitly: all gradients are packed into ONE flat fp64 buffer and reduced with a
single RCCL all-reduce over xGMI (backend "nccl" on ROCm; "gloo" on CPU for the tests) -- one large
message instead of DDP's 25 MB buckets, which is what the point-to-point xGMI links prefer -- then
averaged like DDP does (the reference pre-multiplies lr by n_procs, train.py:136)."""
import torch
import torch.distributed as dist


def allreduce_gradients(params, average=True, group=None):
    """In-place mean (or sum) of .grad over the ranks for every parameter that has a gradient.
    Every rank must hold the same set of gradients (missing ones are treated as zeros)."""
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p.data)
    flat = torch.cat([p.grad.reshape(-1).to(torch.float64) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].reshape(p.grad.shape).to(p.grad.dtype))
        off += n


def allreduce_scalar(t, group=None, op="sum"):
    """Loss / distortion bookkeeping across ranks (the reference does not reduce these at all: every rank
    evaluates the full validation set redundantly, runner.py:124-135)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=group)
    return t


def shard_triplets(triplets, rank, world, epoch=0, seed=0, shuffle=True):
    """This rank's share of a triplet tensor, exactly what DataLoader(sampler=DistributedSampler(...))
    iterates over in train.py:105-110."""
    from sympa_amd.data import distributed_sampler_indices
    idx = distributed_sampler_indices(triplets.shape[0], world, rank, epoch=epoch, seed=seed, shuffle=shuffle)
    return triplets[torch.as_tensor(idx, dtype=torch.long, device=triplets.device)]
