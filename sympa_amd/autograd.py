"""autograd glue for the hot path: forward and backward are both HIP kernels behind the C-ABI.
The backward kernels recompute the forward quantities (nothing but the inputs is saved)."""
import torch

from sympa_amd import ops


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _SiegelDistFn(torch.autograd.Function):
    """manifold.dist(z1, z2) with the analytic backward (sympa_siegel_dist_bwd)."""

    @staticmethod
    def forward(ctx, z1, z2, weights, model, metric):
        ctx.save_for_backward(z1, z2, weights if weights is not None else torch.empty(0, device=z1.device))
        ctx.model, ctx.metric, ctx.has_w = model, metric, weights is not None
        return ops.siegel_dist_forward(z1, z2, model, metric, weights)

    @staticmethod
    def backward(ctx, grad_out):
        z1, z2, w = ctx.saved_tensors
        w = w if ctx.has_w else None
        g1, g2, gw = ops.siegel_dist_backward(z1, z2, grad_out, ctx.model, ctx.metric, w)
        if gw is not None and w is not None:
            gw = gw.reshape(w.shape)
        return g1, g2, (gw if ctx.has_w and ctx.metric == "wsum" else None), None, None


def siegel_dist(z1, z2, model, metric, weights=None):
    if _needs_grad(z1, z2, weights):
        return _SiegelDistFn.apply(z1, z2, weights, model, metric)
    return ops.siegel_dist_forward(z1, z2, model, metric, weights)


class _ModelForwardFn(torch.autograd.Function):
    """Fused Model.forward with the scatter-add backward (sympa_model_backward): the gradient of the
    table is the dense [N,2,n,n] tensor the reference's DDP all-reduces (train.py:59, runner.py:105)."""

    @staticmethod
    def forward(ctx, table, triplets, weights, scale, model, metric, scale_coef):
        dev = table.device
        ctx.save_for_backward(table, triplets, weights if weights is not None else torch.empty(0, device=dev),
                              scale if scale is not None else torch.empty(0, device=dev))
        ctx.cfg = (model, metric, scale_coef, weights is not None, scale is not None)
        return ops.model_forward(table, triplets, model, metric, weights, scale, scale_coef)

    @staticmethod
    def backward(ctx, grad_out):
        table, triplets, w, scale = ctx.saved_tensors
        model, metric, scale_coef, has_w, has_scale = ctx.cfg
        w = w if has_w else None
        scale = scale if has_scale else None
        gt, gw, gs = ops.model_backward(table, triplets, grad_out, model, metric, w, scale, scale_coef)
        if gw is not None and w is not None:
            gw = gw.reshape(w.shape)
        if gs is not None and scale is not None:
            gs = gs.reshape(scale.shape)
        return (gt if ctx.needs_input_grad[0] else None, None,
                gw if (has_w and metric == "wsum" and ctx.needs_input_grad[2]) else None,
                gs if (has_scale and ctx.needs_input_grad[3]) else None, None, None, None)


def model_forward(table, triplets, model, metric, weights=None, scale=None, scale_coef=1.0):
    if _needs_grad(table, weights, scale):
        return _ModelForwardFn.apply(table, triplets, weights, scale, model, metric, scale_coef)
    return ops.model_forward(table, triplets, model, metric, weights, scale, scale_coef)


class _SpdDistFn(torch.autograd.Function):
    """SymmetricPositiveDefinite.dist(x, y) with the analytic backward (sympa_spd_backward_rows)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return ops.spd_dist_forward(x, y)

    @staticmethod
    def backward(ctx, grad_out):
        x, y = ctx.saved_tensors
        rows, _ = ops.spd_backward_rows(x, y, grad_out=grad_out)
        b = x.shape[0]
        return rows[:b], rows[b:]


def spd_dist(x, y):
    if _needs_grad(x, y):
        return _SpdDistFn.apply(x, y)
    return ops.spd_dist_forward(x, y)


class _SpdModelForwardFn(torch.autograd.Function):
    """Model.forward for the spd model: dense [N, n, n] table gradient = per-pair rows + scatter-add kernel."""

    @staticmethod
    def forward(ctx, table, triplets, scale, scale_coef):
        ctx.save_for_backward(table, triplets, scale if scale is not None else torch.empty(0, device=table.device))
        ctx.cfg = (scale_coef, scale is not None)
        return ops.spd_model_forward(table, triplets, scale, scale_coef)

    @staticmethod
    def backward(ctx, grad_out):
        table, triplets, scale = ctx.saved_tensors
        scale_coef, has_scale = ctx.cfg
        scale = scale if has_scale else None
        gs = torch.zeros(1, dtype=torch.float64, device=table.device) if has_scale else None
        rows, _ = ops.spd_backward_rows(table, table, triplets, grad_out=grad_out, scale=scale, scale_coef=scale_coef,
                                        grad_scale=gs)
        gt = torch.zeros_like(table)
        b = triplets.shape[0]
        idx = torch.cat((triplets[:, 0], triplets[:, 1])).contiguous()
        ops.scatter_add_flat_rows_(gt, rows.reshape(2 * b, -1), idx)
        return (gt if ctx.needs_input_grad[0] else None, None,
                gs.reshape(scale.shape) if (has_scale and ctx.needs_input_grad[2]) else None, None)


def spd_model_forward(table, triplets, scale=None, scale_coef=1.0):
    if _needs_grad(table, scale):
        return _SpdModelForwardFn.apply(table, triplets, scale, scale_coef)
    return ops.spd_model_forward(table, triplets, scale, scale_coef)
