"""RiemannianSGD for the Siegel models.

The reference builds `geoopt.optim.RiemannianSGD(params, lr, weight_decay, stabilize=None)` (train.py:66-68);
geoopt is an un-vendored dependency absent from the reference tree, so its step is restated from the
published algorithm (geoopt/optim/rsgd.py, momentum = 0):
    grad <- grad + weight_decay * point;  grad <- manifold.egrad2rgrad(point, grad)
    point <- manifold.retr(point, -lr * grad)
For a table on a Siegel manifold the whole step is ONE HIP kernel over the rows (sympa_rsgd_step);
parameters without a manifold (the model scale, the wsum weights) get the Euclidean update."""
import torch

from sympa_amd import ops
from sympa_amd.manifolds.siegel_manifold import SiegelManifold
from sympa_amd.manifolds.spd import SymmetricPositiveDefinite


class RiemannianSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, weight_decay=0.0, stabilize=None):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))
        self._stabilize = stabilize   # accepted for signature compatibility; Siegel retr already projects
        # Extension: fold torch.nn.utils.clip_grad_norm_(params, max_norm) (runner.py:115) into the step.  The total
        # norm is accumulated on the device and the factor min(1, max_norm / (norm + 1e-6)) is applied to the
        # gradients as the update kernels read them; .grad itself is left unscaled.
        self.clip_max_norm = None
        self._sqnorm = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        sq = coef = None
        if self.clip_max_norm is not None:
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grads and all(t.is_cuda and t.dtype == torch.float64 for t in grads):
                key = str(grads[0].device)
                if key not in self._sqnorm:
                    self._sqnorm[key] = torch.zeros(1, dtype=torch.float64, device=grads[0].device)
                sq = self._sqnorm[key]
                sq.zero_()
                for t in grads:
                    ops.sqnorm_accum_(t if t.is_contiguous() else t.contiguous(), sq)
            else:
                torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"]], self.clip_max_norm)
        for group in self.param_groups:
            lr, wd = group["lr"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                manifold = getattr(p, "manifold", None)
                if isinstance(manifold, SymmetricPositiveDefinite) and p.is_cuda:
                    # geoopt's SPD step: retr(x, -lr x sym(g + wd x) x), one kernel over the table
                    ops.spd_rsgd_step_(p.data, p.grad, lr, wd, clip_sqnorm=sq,
                                       max_norm=self.clip_max_norm if sq is not None else None)
                elif isinstance(manifold, SiegelManifold) and p.is_cuda and sq is not None:
                    ops.rsgd_step_(p.data, p.grad, manifold.model_name, lr, wd, counter=manifold.projected_counter(p.device),
                                   clip_sqnorm=sq, max_norm=self.clip_max_norm)
                elif isinstance(manifold, SiegelManifold) and p.is_cuda:
                    # the kernel adds the number of projected rows to a persistent device counter (read lazily by
                    # manifold.projected_points): no allocation, no synchronisation, graph-capturable
                    ops.rsgd_step_(p.data, p.grad, manifold.model_name, lr, wd, counter=manifold.projected_counter(p.device))
                else:
                    g = p.grad
                    if sq is not None:
                        if coef is None:
                            coef = (self.clip_max_norm / (sq.sqrt() + 1e-6)).clamp(max=1.0)
                        g = g * coef.to(g.dtype)
                    if wd != 0:
                        g = g.add(p, alpha=wd)
                    p.add_(g, alpha=-lr)
        return loss
