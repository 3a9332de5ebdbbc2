"""SiegelManifold: base class of the complex-symmetric-matrix models
(reference sympa/manifolds/siegel_manifold.py:11-167).  Same constructor, attributes and method
signatures; `dist` runs the fused gfx950 kernel instead of ~100 ATen ops with 4 host syncs."""
from abc import ABC
from typing import Optional, Tuple, Union

import torch

from sympa_amd import ops
from sympa_amd.manifolds.base import Manifold
from sympa_amd.manifolds.metrics import Metric, MetricType


def _sym(x):
    return 0.5 * (x + x.transpose(-1, -2))


class SiegelManifold(Manifold, ABC):
    ndim = 1
    reversible = False
    name = "Siegel Space"
    __scaling__ = Manifold.__scaling__.copy()
    model_name = "upper"     # which kernel prologue `dist` uses

    def __init__(self, dims=2, ndim=2, metric=MetricType.RIEMANNIAN, use_xitorch=False):
        super().__init__()
        self.dims = dims
        self.ndim = ndim
        self._projected_points = 0
        self._device_projected = {}       # device -> persistent int32 counter the fused optimiser step adds to
        self.metric = Metric.get(metric, self.dims)

    @property
    def projected_points(self):
        """Number of points projx had to move (runner.py:47-48 logs it).  Counters produced by the fused
        optimiser kernel are folded in here, so the training loop itself never synchronises."""
        for counter in self._device_projected.values():
            self._projected_points += int(counter.item())
            counter.zero_()
        return self._projected_points

    @projected_points.setter
    def projected_points(self, value):
        for counter in self._device_projected.values():
            counter.zero_()
        self._projected_points = value

    def projected_counter(self, device):
        """The device counter sympa_rsgd_step accumulates into.  It persists across steps (nothing is allocated or
        zeroed per step), so the optimiser step can be captured in a hipGraph and replayed."""
        key = str(device)
        if key not in self._device_projected:
            self._device_projected[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._device_projected[key]

    # ------------------------------------------------------------------ the hot path
    def _metric_weights(self):
        return self.metric.weights if self.metric.kind is MetricType.WEIGHTED_SUM else None

    def dist(self, z1: torch.Tensor, z2: torch.Tensor, *, keepdim=False) -> torch.Tensor:
        """z1, z2: b x 2 x n x n on the GPU -> b distances (siegel_manifold.py:41-72)."""
        from sympa_amd.autograd import siegel_dist
        return siegel_dist(z1, z2, self.model_name, self.metric.kind.value, self._metric_weights())

    def vvd(self, z1, z2):
        """Ascending vector-valued distance v (the reference's intermediate, siegel_manifold.py:69-70)."""
        return ops.siegel_dist_forward(z1, z2, self.model_name, "riem", return_vvd=True)[1]

    # ------------------------------------------------------------------ geoopt surface
    def retr(self, x, u):  # siegel_manifold.py:74-87
        return self.projx(x + u)

    def _check_shape(self, shape: Tuple[int], name: str) -> Union[Tuple[bool, Optional[str]], bool]:
        ok = shape[-1] == self.dims and shape[-2] == self.dims   # siegel_manifold.py:89-118
        reason = None if ok else "'{}' on the {} requires more than {} dim".format(name, self, self.dims)
        return ok, reason

    def _check_matrices_are_symmetric(self, x, *, atol=1e-5, rtol=1e-5):  # siegel_manifold.py:120-128
        return torch.allclose(x, x.transpose(-1, -2), atol=atol, rtol=rtol)

    def projx(self, x):  # siegel_manifold.py:130-137
        return torch.stack((_sym(x[:, 0]), _sym(x[:, 1])), dim=1)

    def _projx_kernel(self, z):
        """projx of the concrete model as one HIP kernel; keeps `projected_points` like the reference
        (upper_half.py:64: the reference also synchronises here with `.item()`)."""
        counter = torch.zeros(1, dtype=torch.int32, device=z.device)
        out = ops.projx(z, self.model_name, counter=counter)
        self.projected_points += int(counter.item())
        return out

    def proju(self, x, u):
        return self.egrad2rgrad(x, u)

    def transp(self, x, y, v):  # siegel_manifold.py:142-154: Euclidean transport
        return v

    def expmap(self, x, u):
        pass

    def logmap(self, x, y):
        pass

    def _check_vector_on_tangent(self, x, u, *, atol=1e-5, rtol=1e-5):
        pass
