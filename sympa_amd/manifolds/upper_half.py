"""UpperHalfManifold  H_n = {Z in Sym(n,C) | Im Z > 0}  (reference sympa/manifolds/upper_half.py).

`dist`, `egrad2rgrad`, `projx` (hence `retr`) = HIP kernels.  `inner`, `random`, `_check_point_on_manifold`
are host-side helpers of the reference (init / logging / assertions), kept as torch ops."""
import torch

from sympa_amd import ops
from sympa_amd.manifolds.base import Manifold
from sympa_amd.manifolds.metrics import MetricType
from sympa_amd.manifolds.siegel_manifold import SiegelManifold, _sym


class UpperHalfManifold(SiegelManifold):
    ndim = 1
    reversible = False
    name = "Upper Half Space"
    __scaling__ = Manifold.__scaling__.copy()
    model_name = "upper"

    def __init__(self, dims=2, ndim=2, metric=MetricType.RIEMANNIAN):
        super().__init__(dims=dims, ndim=ndim, metric=metric)

    def egrad2rgrad(self, z, u):  # upper_half.py:25-40: Y G Y on both planes  (HIP kernel)
        return ops.egrad2rgrad(z, u, self.model_name)

    def projx(self, z):  # upper_half.py:42-66 + csym_math.py:252-278  (HIP kernel)
        return self._projx_kernel(z)

    def inner(self, z, u, v=None, *, keepdim=False):  # upper_half.py:68-91: tr[Y^-1 u Y^-1 conj(v)]
        if v is None:
            v = u
        yi = torch.inverse(z[:, 1])
        ur, ui = yi @ u[:, 0] @ yi, yi @ u[:, 1] @ yi
        real = (ur @ v[:, 0] + ui @ v[:, 1]).diagonal(dim1=-2, dim2=-1).sum(-1, keepdim=True).unsqueeze(-1)
        return torch.stack((real, real), dim=1)

    def _check_point_on_manifold(self, z, *, atol=1e-5, rtol=1e-5):  # upper_half.py:93-114
        if not self._check_matrices_are_symmetric(z, atol=atol, rtol=rtol):
            return False, "Matrices are not symmetric"
        ok = bool(torch.det(z[1]) > 0)
        return ok, None if ok else "'x' determinant is not > 0"

    def random(self, *size, dtype=None, device=None, **kwargs):  # upper_half.py:116-131
        from_ = kwargs.get("from_", -0.001)
        to = kwargs.get("to", 0.001)
        n = self.dims
        pert = _sym(torch.empty(size[0], n, n, dtype=torch.get_default_dtype()).uniform_(from_, to))
        imag = torch.eye(n).unsqueeze(0).repeat(size[0], 1, 1) + pert
        real = _sym(torch.empty(size[0], n, n, dtype=torch.get_default_dtype()).uniform_(from_, to))
        return torch.stack((real, imag), dim=1).to(device=device, dtype=dtype)
