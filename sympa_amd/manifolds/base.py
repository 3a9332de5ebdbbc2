"""Minimal geoopt compatibility layer.

The reference subclasses `geoopt.manifolds.base.Manifold` and stores its table in a
`geoopt.ManifoldParameter` (siegel_manifold.py:4,11; embeddings.py:27).  geoopt is an un-vendored
dependency that is not installed here; when it is importable the real classes are used,
otherwise these stand-ins provide exactly the surface the hot path and its callers touch."""
import torch

try:  # pragma: no cover - geoopt is absent in this image
    from geoopt.manifolds.base import Manifold  # type: ignore
    from geoopt import ManifoldParameter  # type: ignore
    HAVE_GEOOPT = True
except Exception:  # noqa: BLE001
    HAVE_GEOOPT = False

    class Manifold(torch.nn.Module):
        """Subset of geoopt.manifolds.base.Manifold used by sympa."""
        __scaling__ = {}
        name = None
        ndim = None
        reversible = None

        def check_point_on_manifold(self, x, *, explain=False, atol=1e-5, rtol=1e-5):
            ok, reason = self._check_shape(x.shape, "x")
            if ok:
                ok, reason = self._check_point_on_manifold(x, atol=atol, rtol=rtol)
            if explain:
                return ok, reason
            return ok

        def assert_check_point_on_manifold(self, x, *, atol=1e-5, rtol=1e-5):
            ok, reason = self.check_point_on_manifold(x, explain=True, atol=atol, rtol=rtol)
            if not ok:
                raise ValueError(f"`x` seems to be a tensor not lying on {self.name} manifold.\nerror: {reason}")

        def retr_transp(self, x, u, v):
            y = self.retr(x, u)
            return y, self.transp(x, y, v)

        def extra_repr(self):
            return ""

        def __repr__(self):
            return f"{self.name} manifold"

    class ManifoldParameter(torch.nn.Parameter):
        """nn.Parameter that remembers the manifold it lives on (geoopt.ManifoldParameter)."""

        def __new__(cls, data=None, manifold=None, requires_grad=True):
            if data is None:
                data = torch.empty(0)
            inst = torch.Tensor._make_subclass(cls, data, requires_grad)
            inst.manifold = manifold
            return inst

        def __repr__(self):
            return f"Parameter on {self.manifold} containing:\n" + torch.Tensor.__repr__(self)

        def __deepcopy__(self, memo):
            # nn.Parameter.__deepcopy__ rebuilds with type(self)(data, requires_grad): the manifold would be lost (and
            # the optimiser would silently take the Euclidean branch for the copy)
            if id(self) in memo:
                return memo[id(self)]
            import copy
            result = type(self)(self.data.clone(memory_format=torch.preserve_format), copy.deepcopy(self.manifold, memo),
                                self.requires_grad)
            memo[id(self)] = result
            return result

        def __reduce_ex__(self, proto):
            return ManifoldParameter, (self.data, self.manifold, self.requires_grad)
