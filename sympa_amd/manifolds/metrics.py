"""Vector-valued-distance metrics (reference sympa/manifolds/metrics.py:6-121).

In the reference each Metric reduces v in R^n to a scalar with torch ops after dist() has built v.
Here the reduction is the epilogue of the HIP kernel; these classes only carry the metric identity
(+ the learnable weights of `wsum`, which must stay a registered Parameter so DDP and the optimiser
see it: SURVEY 8b) and offer `compute_metric` on an explicit v for API compatibility."""
from abc import ABC, abstractmethod
from enum import Enum

import torch


class MetricType(Enum):
    RIEMANNIAN = "riem"
    FINSLER_ONE = "fone"
    FINSLER_INFINITY = "finf"
    FINSLER_MINIMUM = "fmin"
    WEIGHTED_SUM = "wsum"

    @staticmethod
    def from_str(label):
        return {t.value: t for t in MetricType}[label]


class Metric(ABC):
    kind = None   # MetricType

    def __init__(self, dims: int):
        self.dims = dims

    @abstractmethod
    def compute_metric(self, v: torch.Tensor, keepdim=False) -> torch.Tensor:
        raise NotImplementedError

    @classmethod
    def get(cls, type: MetricType, dims: int):
        table = {
            MetricType.RIEMANNIAN: RiemannianMetric,
            MetricType.FINSLER_ONE: FinslerOneMetric,
            MetricType.FINSLER_INFINITY: FinslerInfinityMetric,
            MetricType.FINSLER_MINIMUM: FinslerMinimumEntropyMetric,
            MetricType.WEIGHTED_SUM: FinslerWeightedSumMetric,
        }
        return table[type](dims)


class RiemannianMetric(Metric):
    kind = MetricType.RIEMANNIAN

    def compute_metric(self, v, keepdim=False):
        return torch.norm(v, dim=-1, keepdim=keepdim)


class FinslerOneMetric(Metric):
    kind = MetricType.FINSLER_ONE

    def compute_metric(self, v, keepdim=False):
        return torch.sum(v, dim=-1, keepdim=keepdim)


class FinslerInfinityMetric(Metric):
    kind = MetricType.FINSLER_INFINITY

    def compute_metric(self, v, keepdim=False):
        res = v[:, -1]
        return res.reshape((-1, 1)) if keepdim else res


class FinslerMinimumEntropyMetric(Metric):
    kind = MetricType.FINSLER_MINIMUM

    def __init__(self, dims: int):
        super().__init__(dims)
        self.weights = 2 * (dims + 1 - torch.arange(start=dims + 1, end=1, step=-1).unsqueeze(0))

    def compute_metric(self, v, keepdim=False):
        return torch.sum(self.weights.to(v) * v, dim=-1, keepdim=keepdim)


class FinslerWeightedSumMetric(Metric, torch.nn.Module):
    kind = MetricType.WEIGHTED_SUM

    def __init__(self, dims):
        torch.nn.Module.__init__(self)
        Metric.__init__(self, dims)
        self.weights = torch.nn.parameter.Parameter(torch.ones((1, dims)))

    def compute_metric(self, v, keepdim=False):
        return torch.sum(torch.nn.functional.relu(self.weights) * v, dim=-1, keepdim=keepdim)
