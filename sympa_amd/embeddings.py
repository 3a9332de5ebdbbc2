"""Embedding table + factories for the matrix models (reference sympa/embeddings.py).

Only the Siegel models are in scope: `upper`, `bounded`.  The vector models and
`spd` live in geoopt, not in the reference, and are not part of this hot path; asking for them
raises with that explanation."""
import abc

import torch
import torch.nn as nn

from sympa_amd import config
from sympa_amd.manifolds import BoundedDomainManifold, SymmetricPositiveDefinite, UpperHalfManifold
from sympa_amd.manifolds.base import ManifoldParameter
from sympa_amd.manifolds.metrics import MetricType


class Embeddings(nn.Module, abc.ABC):
    def __init__(self, num_embeddings, dims, manifold, _embeds):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.dims = dims
        self.manifold = manifold
        self.embeds = ManifoldParameter(_embeds, manifold=self.manifold)   # embeddings.py:27

    def forward(self, input_index):   # embeddings.py:29-34 (plain row gather; Model.forward fuses it)
        return self.embeds[input_index]

    def proj_embeds(self):            # embeddings.py:36-39
        with torch.no_grad():
            self.embeds.data = self.manifold.projx(self.embeds.data)
            torch.autograd.graph.increment_version(self.embeds)     # cached packs of the table (ops.PackedTable) key on it

    def check_all_points(self):
        """embeddings.py:41-47 loops over the N rows in Python (a host sync per row, called once per epoch by
        runner.py:180-184).  Same verdict from batched device ops with ONE sync; only when some row fails is
        the reference's per-point check run on that row to produce its (point, reason)."""
        pts = self.embeds.data
        bad = self._first_bad_row(pts)
        if bad is None:
            return True, None, None
        point = pts[bad]
        ok, reason = self.manifold.check_point_on_manifold(point, explain=True)
        return (True, None, None) if ok else (False, point, reason)

    def _first_bad_row(self, pts, atol=1e-5, rtol=1e-5):
        from sympa_amd.manifolds import SymmetricPositiveDefinite, UpperHalfManifold
        diff = (pts - pts.transpose(-1, -2)).abs()
        tol = atol + rtol * pts.transpose(-1, -2).abs()
        flat = (diff > tol).reshape(len(pts), -1).any(dim=1)              # torch.allclose, row by row
        if isinstance(self.manifold, UpperHalfManifold):
            flat |= ~(torch.linalg.det(pts[:, 1]) > 0)                     # upper_half.py:108-113
        elif isinstance(self.manifold, SymmetricPositiveDefinite):
            flat |= ~(torch.linalg.eigvalsh(pts) > -atol).all(dim=-1)
        else:                                                              # bounded_domain.py:141-149
            zc = torch.complex(pts[:, 0], pts[:, 1])
            a = torch.eye(pts.shape[-1], dtype=zc.dtype, device=pts.device) - zc.conj() @ zc
            d = (a - a.conj().transpose(-1, -2)).abs()
            flat |= (d > 1e-8 + 1e-5 * a.abs()).reshape(len(pts), -1).any(dim=1)
        idx = torch.nonzero(flat)
        return int(idx[0]) if idx.numel() else None

    @abc.abstractmethod
    def norm(self):
        pass


class MatrixEmbeddings(Embeddings):
    """Table of complex symmetric matrices: num_embeddings x 2 x dims x dims (embeddings.py:54-74)."""

    def __init__(self, num_embeddings, dims, manifold):
        _embeds = manifold.random(num_embeddings, dims, dims, from_=-config.INIT_EPS, to=config.INIT_EPS)
        if isinstance(manifold, SymmetricPositiveDefinite):   # embeddings.py:70-72: near the identity
            _embeds *= config.INIT_EPS
            _embeds += torch.diag_embed(torch.ones(num_embeddings, dims))
        super().__init__(num_embeddings, dims, manifold, _embeds)

    def norm(self):
        points = self.embeds.data
        return points.reshape(len(points), -1).norm(dim=-1)


class EmbeddingsFactory:
    @classmethod
    def get_embeddings(cls, name: str, num_points: int, dims: int, manifold):
        return cls._get_table(name)(num_embeddings=num_points, dims=dims, manifold=manifold)

    @classmethod
    def _get_table(cls, model_name: str):
        if model_name in ManifoldFactory.sympa_manifolds or model_name == "spd":
            return MatrixEmbeddings
        raise ValueError(f"Unrecognized embedding model for the Siegel hot path: {model_name}")


class ManifoldFactory:
    sympa_manifolds = {"upper": UpperHalfManifold, "bounded": BoundedDomainManifold}   # embeddings.py:145-149
    out_of_scope = {"dual", "euclidean", "poincare", "lorentz", "sphere",
                    "prod-hysph", "prod-hyhy", "prod-hyeu", "prod-sphsph"}

    @classmethod
    def get_manifold(cls, manifold_name, metric_name, dims):
        if manifold_name in cls.out_of_scope:
            raise NotImplementedError(
                f"manifold '{manifold_name}' is outside the MI355X hot path (SURVEY section 2: its "
                "arithmetic lives in geoopt / xitorch, not in the reference)")
        if manifold_name == "spd":      # embeddings.py:142
            return SymmetricPositiveDefinite()
        manifold = cls.sympa_manifolds[manifold_name]
        return manifold(dims=dims, metric=MetricType.from_str(metric_name))
