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
averaged like DDP does (the reference pre-multiplies lr by n_procs, train.py:136).

`GradientExchange` is the persistent-buffer form used by the training harness: gradients are views of one flat
buffer (no per-step `torch.cat`), and for graphs with more nodes than 2 x the global batch the table gradient
travels as touched rows instead of the dense tensor.  `allreduce_gradients` is kept as the one-shot helper."""
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


class GradientExchange:
    """Gradient exchange of data-parallel training with persistent buffers (nothing is allocated or concatenated per
    step, so the step stays stream-ordered and capturable).

    * every gradient lives in ONE flat fp64 buffer allocated here, `p.grad` of each parameter is a view into it;
      `allreduce()` is a single in-place collective on that buffer followed by the 1 / world of DDP's mean;
    * mode "rows" (touched-row exchange): the table gradient does not travel as the dense
      [N, 2, n, n] tensor the reference's DDP all-reduces (train.py:59) but as the 2 b per-pair gradient rows of each
      rank (C-ABI sympa_model_loss_backward_rows) plus their row indices: one all-gather of [2b, 2n^2] rows and one of
      [2b] int64 indices, then every rank scatter-adds all world x 2b rows into its (zeroed) dense gradient
      (sympa_scatter_add_rows, alpha = 1 / world).  Message per step: 2 B 16 n^2 bytes over all ranks (B = global
      batch) instead of N 16 n^2 per rank -- "auto" picks it when that is smaller, i.e. 2 B < N (large graphs with the
      reference's default batch sizes); for BASELINE's configs 2 B > N and the dense all-reduce is the smaller message.
    * mode "sharded" (reduce-scatter -> sharded RSGD + projx -> all-gather): the table gradient is
      reduce-scattered by contiguous row shards (rows padded to a multiple of the world size), every rank runs the
      RiemannianSGD step -- egrad2rgrad, retraction and projection -- on ITS shard of the table only, and the updated rows
      are all-gathered.  Same bytes on the links as the dense all-reduce (which is a reduce-scatter + all-gather of the
      gradient), 1 / world of the optimiser work per GPU; the clip norm costs one 8-byte all-reduce.
    The small parameters (model scale, wsum weights) always go through the flat all-reduce."""

    def __init__(self, params, table=None, local_batch=0, mode="auto", group=None, scatter_fn=None):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.params = [p for p in params if p.requires_grad]
        self.table = table
        if table is not None and not any(p is table for p in self.params):
            raise ValueError("`table` must be one of the parameters")
        n_rows = table.shape[0] if table is not None else 0
        if mode == "auto":
            mode = "rows" if (table is not None and local_batch > 0 and 2 * local_batch * self.world < n_rows) else "dense"
        if mode not in ("dense", "rows", "sharded"):
            raise ValueError(mode)
        if mode == "rows" and (table is None or local_batch <= 0):
            raise ValueError("mode 'rows' needs the table parameter and the per-rank batch size")
        if mode == "sharded" and table is None:
            raise ValueError("mode 'sharded' needs the table parameter")
        self.mode = mode
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        dev = self.params[0].device
        # sharded: the table gradient comes first and is padded to world x rows_per rows (reduce_scatter_tensor wants equal
        # blocks); the padding rows stay zero
        pad = 0
        if mode == "sharded":
            self.rows_per = (n_rows + self.world - 1) // self.world
            pad = (self.rows_per * self.world - n_rows) * table[0].numel()
        # every parameter's gradient view starts on a 256-byte boundary of the flat buffer: the table's 16 n^2-byte gradient rows
        # then sit on cache-line boundaries like a tensor of their own (a one-element scale in front of the table had put
        # every row 8 bytes off: the fp64-atomic scatter touched two lines per 128 bytes, spd n = 16 backward 7.6 -> 9.5 ms)
        ALIGN = 32                                     # doubles

        def up(k):
            return -(-k // ALIGN) * ALIGN
        order = self.params if mode != "sharded" else [table] + [p for p in self.params if p is not table]
        total = sum(up(p.numel() + (pad if p is table else 0)) for p in order)
        self.flat = torch.zeros(total, dtype=torch.float64, device=dev)
        off = 0
        views = {}
        self._small = []           # (offset, numel) of everything but the table, for the "rows" / "sharded" modes
        for p in order:
            if p.dtype != torch.float64:
                raise TypeError("parameters are float64 (reference default dtype, config.py:17-18)")
            v = self.flat[off:off + p.numel()].view(p.shape)
            p.grad = v
            views[id(p)] = v
            if p is table:
                self._table_span = (off, p.numel())
                off += up(p.numel() + pad)             # sharded: the padding rows (and the alignment gap) stay zero
            else:
                self._small.append((off, p.numel()))
                off += up(p.numel())
        self._views = [views[id(p)] for p in self.params]
        self.local_batch = int(local_batch)
        self.small = None
        if mode == "rows":
            rowd = table[0].numel()
            b2 = 2 * self.local_batch
            self.rows = torch.zeros(b2, rowd, dtype=torch.float64, device=dev)
            self.idx = torch.zeros(b2, dtype=torch.int64, device=dev)
            self.rows_all = torch.zeros(self.world * b2, rowd, dtype=torch.float64, device=dev)
            self.idx_all = torch.zeros(self.world * b2, dtype=torch.int64, device=dev)
            # the small parameters share one contiguous tail/head buffer for their all-reduce
            n_small = sum(k for _, k in self._small)
            self.small = torch.zeros(n_small, dtype=torch.float64, device=dev) if n_small else None
        if mode == "sharded":
            rowd = table[0].numel()
            rp = self.rows_per
            self.first_row = self.rank * rp
            self.my_rows = max(0, min(rp, n_rows - self.first_row))          # the last shard may be short (padding)
            self.table_grad_padded = self.flat[:self.world * rp * rowd]                        # reduce-scatter input
            self.shard_grad = torch.zeros((rp,) + tuple(table.shape[1:]), dtype=torch.float64, device=dev)
            self.shard_send = torch.zeros((rp,) + tuple(table.shape[1:]), dtype=torch.float64, device=dev)
            self.table_padded = torch.zeros((self.world * rp,) + tuple(table.shape[1:]), dtype=torch.float64, device=dev)
            # the small gradients are contiguous behind the table gradient: one all-reduce of that tail
            self.small_tail = self.flat[self.world * rp * rowd:]
            self.sq = torch.zeros(1, dtype=torch.float64, device=dev)
        if scatter_fn is None:
            from sympa_amd import ops
            scatter_fn = ops.scatter_add_flat_rows_     # HIP kernel; raises on CPU tensors (no CPU path in the product)
        self.scatter_fn = scatter_fn

    # bytes this rank SENDS per step (ring all-reduce: 2 (G-1)/G of the buffer; all-gather: its own block to G-1 peers)
    @property
    def message_bytes(self):
        g = self.world
        if self.mode in ("dense", "sharded"):     # reduce-scatter + all-gather move what the ring all-reduce moves
            return int(2 * (g - 1) / g * self.flat.numel() * 8)
        small = 0 if self.small is None else int(2 * (g - 1) / g * self.small.numel() * 8)
        return (self.rows.numel() * 8 + self.idx.numel() * 8) * (g - 1) + small

    def zero_(self):
        self.flat.zero_()

    def check_views(self):
        """The parameters' .grad must still be the views created here (zero_grad(set_to_none=True) breaks that)."""
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("a parameter's .grad no longer aliases the exchange buffer; use zero_() / "
                                   "zero_grad(set_to_none=False)")

    def allreduce(self):
        """Dense mode: one in-place all-reduce of the flat buffer, then DDP's mean."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)

    def step_after_exchange(self, opt, max_grad_norm):
        """clip_grad_norm_ + optimizer.step() (runner.py:115-118) on the exchanged gradient (dense / rows modes): every rank
        holds the same mean gradient and takes the same step on its replica of the table."""
        if hasattr(opt, "clip_max_norm"):            # sympa_amd.optim.RiemannianSGD: the clip rides inside the step
            opt.clip_max_norm = max_grad_norm
            try:
                opt.step()
            finally:
                opt.clip_max_norm = None
        else:
            torch.nn.utils.clip_grad_norm_(self.params, max_grad_norm)
            opt.step()

    def _reduce_scatter(self, out, inp):
        if self.world == 1:
            out.view(-1).copy_(inp)
        elif dist.get_backend(self.group) == "gloo":
            # gloo (the CPU / shared-GPU TEST backend) has no reduce-scatter: all-reduce, keep this rank's block
            dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=self.group)
            k = out.numel()
            out.view(-1).copy_(inp[self.rank * k:(self.rank + 1) * k])
        else:
            dist.reduce_scatter_tensor(out.view(-1), inp, op=dist.ReduceOp.SUM, group=self.group)

    def sharded_step(self, opt, max_grad_norm):
        """Sharded mode, after the local backward has accumulated into the flat buffer: reduce-scatter the table gradient,
        all-reduce the small gradients, one 8-byte all-reduce for the clip norm (clip_grad_norm_ over ALL parameters,
        runner.py:115), RiemannianSGD (egrad2rgrad + retr + projx, one kernel) on this rank's rows, plain SGD of the small
        parameters on every rank, all-gather of the updated rows; leaves the gradients zero.  sympa_amd.optim.RiemannianSGD
        on a Siegel table only (the step is issued per shard, not through optimizer.step())."""
        from sympa_amd import ops
        from sympa_amd.manifolds.siegel_manifold import SiegelManifold
        from sympa_amd.manifolds.spd import SymmetricPositiveDefinite
        from sympa_amd.optim import RiemannianSGD
        table = self.table
        manifold = getattr(table, "manifold", None)
        spd = isinstance(manifold, SymmetricPositiveDefinite)
        if not isinstance(opt, RiemannianSGD) or not (spd or isinstance(manifold, SiegelManifold)):
            raise TypeError("sharded_step: sympa_amd.optim.RiemannianSGD on a table of a Siegel or the spd manifold")
        inv = 1.0 / self.world
        self._reduce_scatter(self.shard_grad, self.table_grad_padded)
        if self.small_tail.numel() and self.world > 1:
            dist.all_reduce(self.small_tail, op=dist.ReduceOp.SUM, group=self.group)
        k, r0 = self.my_rows, self.first_row
        if inv != 1.0:                               # DDP's mean
            self.shard_grad.mul_(inv)
            self.small_tail.mul_(inv)
        self.sq.zero_()
        if k > 0:
            ops.sqnorm_accum_(self.shard_grad[:k], self.sq)
        if self.rank == 0 and self.small_tail.numel():
            ops.sqnorm_accum_(self.small_tail, self.sq)
        if self.world > 1:
            dist.all_reduce(self.sq, op=dist.ReduceOp.SUM, group=self.group)
        groups = {id(p): g for g in opt.param_groups for p in g["params"]}
        tg = groups[id(table)]
        if k > 0:
            rows = table.data[r0:r0 + k]
            if spd:          # geoopt's SPD step: retr(x, -lr x sym(g + wd x) x), one kernel over the shard
                ops.spd_rsgd_step_(rows, self.shard_grad[:k], tg["lr"], tg.get("weight_decay", 0.0), clip_sqnorm=self.sq,
                                   max_norm=max_grad_norm)
            else:
                ops.rsgd_step_(rows, self.shard_grad[:k], manifold.model_name, tg["lr"], tg.get("weight_decay", 0.0),
                               counter=manifold.projected_counter(table.device), clip_sqnorm=self.sq, max_norm=max_grad_norm)
            self.shard_send[:k].copy_(rows)
        for p in self.params:
            if p is not table:
                g = groups[id(p)]
                ops.sgd_step_clipped_(p.data, p.grad, g["lr"], g.get("weight_decay", 0.0), clip_sqnorm=self.sq,
                                      max_norm=max_grad_norm)
        if self.world > 1:
            dist.all_gather_into_tensor(self.table_padded, self.shard_send, group=self.group)
            table.data.copy_(self.table_padded[:table.shape[0]])
        ops.table_changed(table)                   # written through .data / raw pointers
        self.flat.zero_()

    def exchange_rows(self, idx_src, idx_dst):
        """Rows mode, after the local backward has written this rank's per-pair rows into `self.rows`
        ([0, b): rows of idx_src, [b, 2b): rows of idx_dst) and accumulated the small gradients into the flat buffer:
        all-gather rows + indices, scatter-add everything into the dense table gradient (mean), all-reduce the rest."""
        b = self.local_batch
        if idx_src.shape[0] != b or idx_dst.shape[0] != b:
            raise ValueError("rows mode exchanges full batches of the size given at construction")
        self.idx[:b].copy_(idx_src)
        self.idx[b:].copy_(idx_dst)
        if self.world > 1:
            dist.all_gather_into_tensor(self.rows_all, self.rows, group=self.group)
            dist.all_gather_into_tensor(self.idx_all, self.idx, group=self.group)
            rows_all, idx_all = self.rows_all, self.idx_all
        else:
            rows_all, idx_all = self.rows, self.idx
        off, k = self._table_span
        tgrad = self.flat[off:off + k].view(self.table.shape)
        tgrad.zero_()
        self.scatter_fn(tgrad, rows_all, idx_all, 1.0 / self.world)
        if self.small is not None and self.world > 1:
            o = 0
            for so_, k_ in self._small:
                self.small[o:o + k_].copy_(self.flat[so_:so_ + k_])
                o += k_
            dist.all_reduce(self.small, op=dist.ReduceOp.SUM, group=self.group)
            o = 0
            for so_, k_ in self._small:
                self.flat[so_:so_ + k_].copy_(self.small[o:o + k_]).mul_(1.0 / self.world)
                o += k_


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
