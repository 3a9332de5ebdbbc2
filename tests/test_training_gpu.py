"""GPU end-to-end: the build's own training harness (fused loss+backward kernel, gradient clip, fused RSGD
kernel) actually embeds a graph: average distortion of the 125-node 3D grid (BASELINE.json configs[0]) drops."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("manifold,metric", [("upper", "riem"), ("bounded", "fone")])
def test_embedding_a_grid_reduces_distortion(manifold, metric):
    import train_siegel
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", manifold, "--metric", metric,
                                             "--dims", "2", "--epochs", "40", "--batch_size", "512",
                                             "--val_every", "10", "--learning_rate", "0.02", "--burnin", "5"])
    model, hist = train_siegel.train(args, log=lambda *_: None)
    first, last = hist[0][2], hist[-1][2]
    assert last < 0.6 * first and last < 0.5, hist
    ok, point, reason = model.check_all_points()
    assert ok, reason


def test_embedding_at_dims_8_trains(tmp_path):
    """configs[3] trains at n = 8: the n = 8 forward, fused loss+backward and RSGD kernels together."""
    import train_siegel
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", "upper", "--metric", "riem",
                                             "--dims", "8", "--epochs", "12", "--batch_size", "512",
                                             "--val_every", "4", "--learning_rate", "0.02", "--burnin", "3"])
    model, hist = train_siegel.train(args, log=lambda *_: None)
    first, last = hist[0][2], hist[-1][2]
    assert last < 0.8 * first, hist
    ok, point, reason = model.check_all_points()
    assert ok, reason
