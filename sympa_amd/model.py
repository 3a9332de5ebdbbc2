"""Graph-embedding model on a Siegel manifold (reference sympa/model.py:8-47).

`forward` is the drop-in boundary of the hot path: instead of two `index` gathers that materialise
2 x [b,2,n,n] in HBM followed by ~100 ATen ops, one fused HIP kernel reads the two table rows of
each pair straight from the table and writes the b scaled distances."""
import torch
import torch.nn as nn

from sympa_amd import autograd as sa
from sympa_amd.embeddings import EmbeddingsFactory, ManifoldFactory
from sympa_amd.manifolds.metrics import MetricType


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

    def forward(self, input_triplet):
        """input_triplet: int64 [b, 2|3] (src_id, dst_id[, graph_distance]) -> b distances * scale
        (model.py:16-30)."""
        man = self.manifold
        if man.model_name == "spd":
            return sa.spd_model_forward(self.embeddings.embeds, input_triplet, self.scale, self.scale_coef)
        weights = man.metric.weights if man.metric.kind is MetricType.WEIGHTED_SUM else None
        return sa.model_forward(self.embeddings.embeds, input_triplet, man.model_name, man.metric.kind.value,
                                weights, self.scale, self.scale_coef)

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
