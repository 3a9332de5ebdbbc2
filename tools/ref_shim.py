"""Probe shim that lets the *reference* hot-path modules import in this container.

Used ONLY by tools/make_golden.py (fixture generation) and by ad-hoc validation here.
Nothing in the product, tests, bench or smoke imports this file; /root/reference does
not exist on the GPU box.

What is stubbed:
  * geoopt.linalg.batch_linalg.sym       -> 0.5 * (x + x^T)   (only geoopt fn sympa.math uses)
  * geoopt.manifolds.base.Manifold       -> nn.Module with check_point_on_manifold
  * torch.symeig(y, eigenvectors=...)    -> torch.linalg.eigh(y, UPLO='U')  (old default upper=True)
"""
import sys
import types
import torch

REFERENCE_ROOT = "/root/reference"


def install():
    if "geoopt" in sys.modules and getattr(sys.modules["geoopt"], "_sympa_probe_shim", False):
        return
    geoopt = types.ModuleType("geoopt")
    geoopt._sympa_probe_shim = True
    linalg = types.ModuleType("geoopt.linalg")
    batch_linalg = types.ModuleType("geoopt.linalg.batch_linalg")

    def sym(x):
        return 0.5 * (x.transpose(-1, -2) + x)

    batch_linalg.sym = sym
    linalg.batch_linalg = batch_linalg
    manifolds = types.ModuleType("geoopt.manifolds")
    base = types.ModuleType("geoopt.manifolds.base")

    class Manifold(torch.nn.Module):
        __scaling__ = {}
        ndim = 0
        name = "shim"
        reversible = False

        def check_point_on_manifold(self, x, *, explain=False, atol=1e-5, rtol=1e-5):
            ok, reason = self._check_shape(x.shape, "x")
            if ok:
                ok, reason = self._check_point_on_manifold(x, atol=atol, rtol=rtol)
            if explain:
                return ok, reason
            return ok

    base.Manifold = Manifold
    manifolds.base = base
    geoopt.linalg = linalg
    geoopt.manifolds = manifolds
    sys.modules["geoopt"] = geoopt
    sys.modules["geoopt.linalg"] = linalg
    sys.modules["geoopt.linalg.batch_linalg"] = batch_linalg
    sys.modules["geoopt.manifolds"] = manifolds
    sys.modules["geoopt.manifolds.base"] = base

    if not hasattr(torch, "_sympa_orig_symeig"):
        torch._sympa_orig_symeig = getattr(torch, "symeig", None)

        def symeig(y, eigenvectors=False, upper=True):
            w, v = torch.linalg.eigh(y, UPLO="U" if upper else "L")
            return w, v

        torch.symeig = symeig

    # the reference's manifolds/__init__ imports compact_dual which needs xitorch lazily only
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Returns (sm, cayley module, takagi module, UpperHalfManifold, BoundedDomainManifold, metrics module)."""
    install()
    import sympa.math.csym_math as sm
    import sympa.math.cayley_transform as cayley
    import sympa.math.takagi_factorization as takagi
    from sympa.manifolds.upper_half import UpperHalfManifold
    from sympa.manifolds.bounded_domain import BoundedDomainManifold
    import sympa.manifolds.metrics as metrics
    return sm, cayley, takagi, UpperHalfManifold, BoundedDomainManifold, metrics
