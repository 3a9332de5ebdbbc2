"""SymmetricPositiveDefinite: the `spd` model of the reference (sympa/embeddings.py:6,70-72,142).

In the reference this class IS geoopt.manifolds.SymmetricPositiveDefinite() (default affine-invariant metric):
none of its arithmetic is in the reference tree, geoopt is not installed here, and no reference test touches
it -- parity is UNPINNED with respect to geoopt: the oracle restates geoopt's published formulas, and the
kernels are checked against that restatement and against 50-digit mpmath evaluations of the same formulas
(tests/golden/spd_n*.npz).  dist (forward and backward), egrad2rgrad, retr, projx are HIP kernels."""
import torch

from sympa_amd import ops
from sympa_amd.manifolds.base import Manifold


def _sym(x):
    return 0.5 * (x + x.transpose(-1, -2))


class SymmetricPositiveDefinite(Manifold):
    name = "SymmetricPositiveDefinite"
    ndim = 2
    reversible = False
    __scaling__ = Manifold.__scaling__.copy()
    model_name = "spd"

    def __init__(self, default_metric="AIM"):
        super().__init__()
        if str(default_metric).upper() not in ("AIM", "SPDMETRIC.AIM"):
            raise NotImplementedError("only geoopt's default affine-invariant metric (AIM) is built")
        self.projected_points = 0

    def dist(self, x, y, *, keepdim=False):
        """|| log(x^-1/2 y x^-1/2) ||_F  (geoopt SymmetricPositiveDefinite.dist, AIM); differentiable."""
        from sympa_amd import autograd as sa
        d = sa.spd_dist(x, y)
        return d.unsqueeze(-1).unsqueeze(-1) if keepdim else d

    def egrad2rgrad(self, x, u):
        """geoopt: x @ sym(u) @ x."""
        return ops.spd_egrad2rgrad(x, u)

    def proju(self, x, u):
        return _sym(u)

    def retr(self, x, u):
        """geoopt: sym(x + u + 1/2 u x^-1 u), evaluated by the step kernel with lr = -1 on egrad-free input: host-side
        torch here (init / tests); the optimiser uses the fused sympa_spd_rsgd_step."""
        return _sym(x + u + 0.5 * u @ torch.linalg.inv(x) @ u)

    def _check_shape(self, shape, name):
        ok = len(shape) >= 2 and shape[-1] == shape[-2]
        return ok, None if ok else f"`{name}` should be a square matrix"

    def _check_point_on_manifold(self, x, *, atol=1e-5, rtol=1e-5):
        if not torch.allclose(x, x.transpose(-1, -2), atol=atol, rtol=rtol):
            return False, "`x != x.transpose` with atol={}, rtol={}".format(atol, rtol)
        ok = bool((torch.linalg.eigvalsh(x) > -atol).all())
        return ok, None if ok else "eigenvalues of x are not all greater than 0."

    def projx(self, x):
        """geoopt: sym_funcm(sym(x), abs) -- HIP kernel for device tensors, torch on the host (initialisation)."""
        if x.is_cuda:
            return ops.spd_projx(x)
        s = _sym(x)
        lam, v = torch.linalg.eigh(s)
        return v @ torch.diag_embed(lam.abs()) @ v.transpose(-1, -2)

    def random(self, *size, dtype=None, device=None, **kwargs):
        """geoopt SymmetricPositiveDefinite.random: expm(sym(0.5 * randn))."""
        t = _sym(0.5 * torch.randn(*size, dtype=torch.get_default_dtype()))
        lam, v = torch.linalg.eigh(t)
        return (v @ torch.diag_embed(torch.exp(lam)) @ v.transpose(-1, -2)).to(device=device, dtype=dtype)
