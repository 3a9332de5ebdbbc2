"""autograd glue for the hot path.  Forward = HIP kernel through the C-ABI.  The analytic backward
kernel is SURVEY 8f-1 (next row); until it lands, differentiating through dist raises instead of
silently falling back to torch ops."""
import torch

from sympa_amd import ops


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _SiegelDistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, weights, model, metric):
        ctx.save_for_backward(z1, z2, weights if weights is not None else torch.empty(0))
        ctx.model, ctx.metric = model, metric
        return ops.siegel_dist_forward(z1, z2, model, metric, weights)

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError(
            "backward of the Siegel distance kernel is not built yet (SURVEY 8f-1); "
            "run forward under torch.no_grad()")


def siegel_dist(z1, z2, model, metric, weights=None):
    if _needs_grad(z1, z2, weights):
        return _SiegelDistFn.apply(z1, z2, weights, model, metric)
    return ops.siegel_dist_forward(z1, z2, model, metric, weights)


class _ModelForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, triplets, weights, scale, model, metric, scale_coef):
        ctx.save_for_backward(table, triplets)
        return ops.model_forward(table, triplets, model, metric, weights, scale, scale_coef)

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError(
            "backward of the fused Model.forward kernel is not built yet (SURVEY 8f-1); "
            "run forward under torch.no_grad()")


def model_forward(table, triplets, model, metric, weights=None, scale=None, scale_coef=1.0):
    if _needs_grad(table, weights, scale):
        return _ModelForwardFn.apply(table, triplets, weights, scale, model, metric, scale_coef)
    return ops.model_forward(table, triplets, model, metric, weights, scale, scale_coef)
