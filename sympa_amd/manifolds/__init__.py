from sympa_amd.manifolds.siegel_manifold import SiegelManifold
from sympa_amd.manifolds.upper_half import UpperHalfManifold
from sympa_amd.manifolds.bounded_domain import BoundedDomainManifold
from sympa_amd.manifolds.spd import SymmetricPositiveDefinite
from sympa_amd.manifolds.metrics import MetricType, Metric

__all__ = ["SiegelManifold", "UpperHalfManifold", "BoundedDomainManifold", "SymmetricPositiveDefinite", "MetricType", "Metric"]
