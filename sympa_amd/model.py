"""Graph-embedding model on a Siegel manifold (reference sympa/model.py:8-47).

`forward` is the drop-in boundary of the hot path: instead of two `index` gathers that materialise
2 x [b,2,n,n] in HBM followed by ~100 ATen ops, one fused HIP kernel reads the two table rows of
each pair straight from the table and writes the b scaled distances."""
import os

import torch
import torch.nn as nn

from sympa_amd import autograd as sa
from sympa_amd import ops
from sympa_amd.embeddings import EmbeddingsFactory, ManifoldFactory
from sympa_amd.manifolds.metrics import MetricType


# forward() under no_grad takes the packed path (two launches) from this many pairs per call on
PACKED_MIN_PAIRS = 4096


class _SpdBatches:
    """forward_batches for the spd model: one launch per batch (its kernel is already many waves per SIMD deep at
    the batch sizes it is used with; no multi-batch kernel)."""

    def __init__(self, model, batches, outs):
        self.model, self.batches, self.outs = model, batches, outs

    def set_streams(self, streams):
        pass

    def run(self):
        from sympa_amd import ops
        m = self.model
        pk = m.packed_table()
        if pk is not None and sum(int(t.shape[0]) for t in self.batches) >= PACKED_MIN_PAIRS:
            pk.ensure(m.embeddings.embeds)          # (key comparison + the device-side digest check: ops.PackedTable)
            for t, o in zip(self.batches, self.outs):
                ops.spd_model_forward_packed(pk, t, m.scale.data, m.scale_coef, out=o)
            return
        for t, o in zip(self.batches, self.outs):
            ops.spd_model_forward(m.embeddings.embeds.data, t, m.scale.data, m.scale_coef, out=o)


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.manifold = ManifoldFactory.get_manifold(manifold_name=args.manifold, metric_name=args.metric,
                                                     dims=args.dims)
        self.embeddings = EmbeddingsFactory.get_embeddings(args.manifold, args.num_points, args.dims,
                                                           self.manifold)
        self.scale_coef = args.scale_coef
        self.scale = torch.nn.Parameter(torch.Tensor([self.scale_coef * args.scale_init]),
                                        requires_grad=args.train_scale)

    def _forward_cache(self):
        """(table, scale, wsum weights or None, model name, metric name, + the owners the entries were read from): the
        attribute chains of forward() resolved once -- nn.Module attribute lookups cost ~0.5 us each and a forward call is a
        7 us kernel.  `.to(device)` and `embeds.data = ...` keep the Parameter objects; REPLACING one (`model.embeddings.embeds =
        nn.Parameter(...)`, a new manifold or metric module) is seen by `_forward_cache_valid` (plain dict lookups on the owning
        modules, ~0.2 us together) and the tuple is rebuilt, so forward() never runs on stale tensors while forward_batches /
        distortion read the live attributes."""
        man = self.manifold
        emb = self.embeddings
        spd = man.model_name == "spd"
        metric = None if spd else man.metric
        wsum = (not spd) and metric.kind is MetricType.WEIGHTED_SUM
        c = (emb.embeds, self.scale, metric.weights if wsum else None, man.model_name,
             None if spd else metric.kind.value, emb, man, metric)
        self.__dict__["_fwd"] = c
        return c

    def _forward_cache_valid(self, c):
        emb, man, metric = c[5], c[6], c[7]
        mods = self._modules
        if mods.get("embeddings") is not emb or mods.get("manifold") is not man or self._parameters.get("scale") is not c[1]:
            return False
        if emb._parameters.get("embeds") is not c[0]:
            return False
        if metric is not None:
            if man._modules.get("metric", man.__dict__.get("metric")) is not metric or metric.kind.value != c[4]:
                return False
            if c[2] is not None and metric._parameters.get("weights") is not c[2]:
                return False
        return True

    def forward(self, input_triplet):
        """input_triplet: int64 [b, 2|3] (src_id, dst_id[, graph_distance]) -> b distances * scale
        (model.py:16-30)."""
        c = self.__dict__.get("_fwd")
        if c is None or not self._forward_cache_valid(c):
            c = self._forward_cache()
        table, scale, weights, model_name, metric_name = c[:5]
        if model_name == "spd":
            if not (torch.is_grad_enabled() and (table.requires_grad or scale.requires_grad)) and \
                    input_triplet.shape[0] >= PACKED_MIN_PAIRS:
                pk = self.packed_table()            # no autograd graph to build: the packed table once its version is seen twice
                if pk is not None and pk.current(table):
                    return ops.spd_model_forward_packed(pk, input_triplet, scale, self.scale_coef)
            return sa.spd_model_forward(table, input_triplet, scale, self.scale_coef)
        if torch.is_grad_enabled() and (table.requires_grad or scale.requires_grad or
                                        (weights is not None and weights.requires_grad)):
            return sa.model_forward(table, input_triplet, model_name, metric_name, weights, scale, self.scale_coef)
        # no autograd graph to build (Runner.evaluate / build_distance_matrix run under no_grad): straight to the binding --
        # dims 5..8 over the packed table once the same table version is seen a second time (ops.PackedTable).  A SINGLE call pays
        # the pack's validity check (a device-side digest of the table: ops.PackedTable.strict) every time: for the upper model that
        # is what the pack saves (n = 8, 262 144 pairs of 45 500 rows: packed 121 + check 18-21 us against 133 dense; n = 6: 75 + 14
        # against 84), so its single calls stay on the dense kernels and the pack serves the list forms (one check per K batches);
        # the bounded model gains far more than the check costs (154 + 18 against 244).
        if 5 <= table.shape[-1] <= 8 and input_triplet.shape[0] >= PACKED_MIN_PAIRS:
            pk = self.packed_table()
            if pk is not None and (model_name == "bounded" or not pk.strict) and pk.current(table, input_triplet.shape[0]):
                return ops.model_forward_packed(pk, input_triplet, metric_name, weights, scale, self.scale_coef)
        return ops.model_forward(table, input_triplet, model_name, metric_name, weights, scale, self.scale_coef)

    def packed_table(self):
        """The ops.PackedTable of this model's embedding table (dims 5..8 of the Siegel models, n in ops.SPD_PACKED_DIMS of the spd
        model, on the GPU; None elsewhere, with `model.use_packed = False` or SYMPA_NO_PACKED=1, or while the self-check has demoted
        the spd forward of these dims): upper triangles + inverted Cholesky factor per point, made once per table state and shared by
        forward() under no_grad, forward_batches() and evaluate().  Whether the pack still is the table's image is decided by the
        host key (storage, shape, torch version counter) AND, unless `packed_table().strict = False`, by a digest of the table's bytes
        computed on the device in front of every use (one read of the table, no synchronisation): `.data` writes that move no version
        counter (embeddings.py:36-39, torch-1.5 optimisers) are seen.  Costs a second table-sized device buffer while it lives
        (`packed_table().invalidate(release=True)` frees it)."""
        if not self.__dict__.get("use_packed", True) or os.environ.get("SYMPA_NO_PACKED"):
            return None
        table = self.embeddings.embeds
        name = self.manifold.model_name
        cls = ops.SpdPackedTable if name == "spd" else ops.PackedTable
        if not cls.supported(table, name):
            return None
        pk = self.__dict__.get("_packed")
        if pk is None or pk.model != name:
            pk = self.__dict__["_packed"] = cls(name)
        return pk

    def _metric_key(self):
        """What a cached plan bakes in of the metric: its kind and the address of the wsum weights."""
        man = self.manifold
        if man.model_name == "spd":
            return None
        k = man.metric.kind
        return (k.value, man.metric.weights.data_ptr() if k is MetricType.WEIGHTED_SUM else 0)

    # ---- lists of batches: the consumer is Runner.evaluate's loop (runner.py:124-135), one forward() per batch ----
    def prepare_batches(self, batches, outs=None):
        """Validates a list of int64 [b_i, 2|3] batches once, allocates one output per batch (or takes `outs`) and
        builds the host-side pointer arrays of C-ABI sympa_model_forward_batches.  The returned plan is what
        `forward_batches` runs; it keeps the tensors alive and reads table, scale and metric weights through their
        device pointers, so parameter updates in place are seen, while `model.to(...)` needs a new plan."""
        from sympa_amd import ops
        man = self.manifold
        table = self.embeddings.embeds.data
        batches = list(batches)
        if outs is None:
            flat = torch.empty(sum(int(t.shape[0]) for t in batches), dtype=torch.float64, device=table.device)
            outs, s = [], 0
            for t in batches:
                outs.append(flat[s:s + t.shape[0]])
                s += t.shape[0]
        outs = list(outs)
        if man.model_name == "spd":
            return _SpdBatches(self, batches, outs)
        weights = man.metric.weights if man.metric.kind is MetricType.WEIGHTED_SUM else None
        pk = self.packed_table()
        if pk is not None and sum(int(t.shape[0]) for t in batches) >= PACKED_MIN_PAIRS:
            # dims 5..8: the list goes over the packed table (repacked by the plan's run() whenever the table's version moved)
            plan = ops.PackedBatchedForward(pk, self.embeddings.embeds, batches, outs, man.metric.kind.value, weights,
                                            self.scale.data, self.scale_coef)
        else:
            plan = ops.BatchedForward(table, batches, outs, man.model_name, man.metric.kind.value, weights, self.scale.data,
                                      self.scale_coef, flags=ops.FLAG_FUSE)
        plan.outs = outs
        return plan

    def forward_batches(self, batches, outs=None):
        """forward() over a list of batches (no autograd): ONE C call; for dims <= 8 the batches go 32 at a time
        through the fused multi-batch kernel (`siegel_dist_multi_kernel`), which is what fills the chip when a batch is
        one wave per SIMD or less.  `batches` is a list of int64 [b_i, 2|3] tensors or a plan from `prepare_batches`;
        for a list the plan is built on first use and cached (keyed by the identity of the tensors, which the plan keeps
        alive; at most 8 plans).  Returns the list of [b_i] outputs, enqueued on the current stream.  With outs=None the
        outputs belong to the cached plan: a second call on the SAME list object writes into the same tensors (clone what
        must survive the next call, or pass `outs`)."""
        if isinstance(batches, (list, tuple)):
            table = self.embeddings.embeds
            key = (tuple(map(id, batches)), None if outs is None else tuple(map(id, outs)),
                   table.data_ptr(), self.scale.data_ptr(), self._metric_key())
            cache = self.__dict__.setdefault("_batch_plans", {})
            plan = cache.get(key)
            if plan is None:
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                plan = cache[key] = self.prepare_batches(batches, outs)
        else:
            plan = batches
        plan.set_streams(None)       # torch's current stream
        plan.run()
        return plan.outs

    def evaluate(self, src_dst_ids, graph_distances, batch_size):
        """Runner.evaluate (runner.py:124-135): average distortion |d_manifold - d_graph| / d_graph (metrics.py:21) over
        every triplet, the model called once per `batch_size` triplets.  Here the batches are row slices of
        `src_dst_ids` [T, 2|3] handed over as one list; the plan and the [T] output are cached on the model (the
        evaluation split is the same every epoch).  One host sync (the returned float)."""
        ids = src_dst_ids if src_dst_ids.is_contiguous() else src_dst_ids.contiguous()
        total = ids.shape[0]
        if total == 0:
            raise ValueError("evaluate() over an empty split (statistics.mean raises in the reference too)")
        table = self.embeddings.embeds
        key = (ids.data_ptr(), total, int(batch_size), table.data_ptr(), self.scale.data_ptr(), self._metric_key())
        ev = self.__dict__.get("_eval_plan")
        if ev is None or ev[0] != key:
            out = torch.empty(total, dtype=torch.float64, device=table.device)
            batches = [ids[s:s + batch_size] for s in range(0, total, batch_size)]
            outs = [out[s:s + batch_size] for s in range(0, total, batch_size)]
            ev = (key, self.prepare_batches(batches, outs), out, ids)
            self.__dict__["_eval_plan"] = ev
        with torch.no_grad():
            self.forward_batches(ev[1])
            # the reference hands graph distances over as [b, 1] (metrics.py:21 docstring): flatten, never broadcast
            gd = graph_distances.to(device=table.device, dtype=torch.float64).reshape(-1)
            if gd.numel() != total:
                raise ValueError(f"{total} triplets but {gd.numel()} graph distances")
            return float(((ev[2] - gd).abs() / gd).sum()) / total

    def fused_loss_backward(self, input_triplet, graph_distances, loss_scale=1.0, loss_out=None, zero_loss_out=True):
        """Extension (not in the reference API): the body of Runner.train_epoch's inner loop
        (runner.py:101-105: forward, AverageDistortionLoss / grad_accum_steps, loss.backward()) as ONE
        HIP kernel.  Accumulates into the parameters' .grad exactly like loss.backward() and returns the
        loss as a 1-element device tensor (no host sync); `loss_out` (1-element fp64 device tensor) is zeroed
        (unless zero_loss_out=False: the caller has) and used for it when given (a persistent buffer: fewer graph nodes per step)."""
        from sympa_amd import ops
        man = self.manifold
        table = self.embeddings.embeds
        dev = table.device
        if table.grad is None:
            table.grad = torch.zeros_like(table.data)
        if man.model_name == "spd":
            # n <= 2: two launches, per-pair gradient rows (backward + loss fused), then the coalesced scatter-add
            gs = None
            if self.scale.requires_grad:
                if self.scale.grad is None:
                    self.scale.grad = torch.zeros_like(self.scale.data)
                gs = self.scale.grad
            loss = torch.zeros(1, dtype=torch.float64, device=dev) if loss_out is None else (loss_out.zero_() if zero_loss_out else loss_out)
            b = input_triplet.shape[0]
            n = table.shape[-1]
            if n >= 3 and table.grad.is_contiguous():
                # one launch: loss, backward and the scatter into .grad (sympa_spd_loss_backward)
                ops.spd_loss_backward(table.data, input_triplet, table.grad, graph_dist=graph_distances, scale=self.scale.data,
                                      scale_coef=self.scale_coef, loss_scale=loss_scale, loss=loss, grad_scale=gs)
                return loss
            ws = getattr(self, "_spd_rows", None)
            if ws is None or ws.shape[0] < 2 * b or ws.device != dev:
                ws = torch.empty(2 * b, n, n, dtype=torch.float64, device=dev)
                self._spd_rows = ws
            ops.spd_backward_rows(table.data, table.data, input_triplet, graph_dist=graph_distances, scale=self.scale.data,
                                  scale_coef=self.scale_coef, loss_scale=loss_scale, loss=loss, grad_scale=gs, rows=ws)
            idx = torch.cat((input_triplet[:, 0], input_triplet[:, 1])).contiguous()
            ops.scatter_add_flat_rows_(table.grad, ws[:2 * b].reshape(2 * b, -1), idx)
            return loss
        wsum = man.metric.kind is MetricType.WEIGHTED_SUM
        weights = gw = None
        if wsum:
            weights = man.metric.weights
            if weights.grad is None:
                weights.grad = torch.zeros_like(weights.data)
            gw = weights.grad
        gs = None
        if self.scale.requires_grad:
            if self.scale.grad is None:
                self.scale.grad = torch.zeros_like(self.scale.data)
            gs = self.scale.grad
        loss = torch.zeros(1, dtype=torch.float64, device=dev) if loss_out is None else (loss_out.zero_() if zero_loss_out else loss_out)
        ops.model_loss_backward(table.data, input_triplet, graph_distances, table.grad, loss, man.model_name,
                                man.metric.kind.value, None if weights is None else weights.data, gw, self.scale.data,
                                gs, self.scale_coef, loss_scale)
        return loss

    def fused_loss_backward_rows(self, input_triplet, graph_distances, grad_rows, loss_scale=1.0):
        """fused_loss_backward with the table gradient left as per-pair rows in `grad_rows` [2b, 2, n, n] (written;
        rows [0, b) belong to the src ids, [b, 2b) to the dst ids) for the touched-row gradient exchange of
        sympa_amd.distributed.GradientExchange; the scale / wsum-weight gradients accumulate into .grad as usual."""
        from sympa_amd import ops
        man = self.manifold
        table = self.embeddings.embeds
        if man.model_name == "spd":
            gs = None
            if self.scale.requires_grad:
                if self.scale.grad is None:
                    self.scale.grad = torch.zeros_like(self.scale.data)
                gs = self.scale.grad
            loss = torch.zeros(1, dtype=torch.float64, device=table.device)
            ops.spd_backward_rows(table.data, table.data, input_triplet, graph_dist=graph_distances, scale=self.scale.data,
                                  scale_coef=self.scale_coef, loss_scale=loss_scale, loss=loss, grad_scale=gs,
                                  rows=grad_rows.view(-1, table.shape[-1], table.shape[-1]))
            return loss
        wsum = man.metric.kind is MetricType.WEIGHTED_SUM
        weights = gw = None
        if wsum:
            weights = man.metric.weights
            if weights.grad is None:
                weights.grad = torch.zeros_like(weights.data)
            gw = weights.grad
        gs = None
        if self.scale.requires_grad:
            if self.scale.grad is None:
                self.scale.grad = torch.zeros_like(self.scale.data)
            gs = self.scale.grad
        loss = torch.zeros(1, dtype=torch.float64, device=table.device)
        ops.model_loss_backward_rows(table.data, input_triplet, graph_distances, grad_rows, loss, man.model_name,
                                     man.metric.kind.value, None if weights is None else weights.data, gw,
                                     self.scale.data, gs, self.scale_coef, loss_scale)
        return loss

    def distance_matrix(self, row_begin=0, row_count=None):
        """Extension: the N x N matrix Runner.build_distance_matrix (runner.py:142-154) assembles with N
        forward calls, as one launch (or one launch per row block).  Diagonal exactly 0."""
        from sympa_amd import ops
        man = self.manifold
        if man.model_name == "spd":
            # the same pairs (i, j) the reference's loop feeds to forward(), through the spd kernel
            emb = self.embeddings.embeds
            n_pts = emb.shape[0]
            row_count = n_pts - row_begin if row_count is None else row_count
            rows = torch.arange(row_begin, row_begin + row_count, device=emb.device)
            cols = torch.arange(n_pts, device=emb.device)
            pairs = torch.stack((rows.repeat_interleave(n_pts), cols.repeat(row_count)), 1)
            out = ops.spd_model_forward(emb, pairs, self.scale, self.scale_coef).reshape(row_count, n_pts)
            out[torch.arange(row_count, device=emb.device), rows] = 0.0      # runner.py:152
            return out
        weights = man.metric.weights if man.metric.kind is MetricType.WEIGHTED_SUM else None
        return ops.all_pairs_dist(self.embeddings.embeds, man.model_name, man.metric.kind.value, weights, self.scale,
                                  self.scale_coef, row_begin, row_count)

    def distance(self, src_embeds, dst_embeds):   # model.py:32-38
        return self.manifold.dist(src_embeds, dst_embeds)

    def get_scale(self):                          # model.py:40-41
        return (self.scale / self.scale_coef).clamp_min(0.1)

    def check_all_points(self):
        return self.embeddings.check_all_points()

    def embeds_norm(self):
        return self.embeddings.norm()
