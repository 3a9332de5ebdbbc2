"""GPU end-to-end: the build's own training harness (fused loss+backward kernel, gradient clip, fused RSGD
kernel) actually embeds a graph: average distortion of the 125-node 3D grid (BASELINE.json configs[0]) drops."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("manifold,metric,dims", [("upper", "riem", 2), ("bounded", "fone", 2), ("bounded", "finf", 4)])
def test_embedding_a_grid_reduces_distortion(manifold, metric, dims):
    """configs[0] (grid, n = 2), and configs[2]'s model (bounded, F-infinity metric, n = 4) on the same small graph."""
    import train_siegel
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", manifold, "--metric", metric,
                                             "--dims", str(dims), "--epochs", "40", "--batch_size", "512",
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


@pytest.mark.parametrize("dims", [3, 10])
def test_graphed_step_equals_eager_step(dims):
    """dims 10: the sixteen-lanes kernels, including the memset + gated projection of the RSGD step, inside the graph.
    One hipGraph replay per batch (sympa_amd/train_step.py) trains exactly like the kernel-by-kernel step: same
    distortion history (fp64 atomics make the last bits of a gradient order-dependent, hence a tolerance)."""
    import train_siegel
    common = ["--graph", "grid3d-125", "--manifold", "upper", "--metric", "riem", "--dims", str(dims), "--epochs", "12",
              "--batch_size", "512", "--val_every", "3", "--learning_rate", "0.02", "--burnin", "4"]
    _, h_graph = train_siegel.train(train_siegel.parser().parse_args(common), log=lambda *_: None)
    _, h_eager = train_siegel.train(train_siegel.parser().parse_args(common + ["--no_graph_step"]), log=lambda *_: None)
    assert len(h_graph) == len(h_eager) == 4
    for a, b in zip(h_graph, h_eager):
        assert a[0] == b[0]
        assert abs(a[1] - b[1]) < 1e-8 * abs(b[1]) and abs(a[2] - b[2]) < 1e-8 * abs(b[2]), (a, b)
    assert h_graph[-1][2] < 0.9 * h_graph[0][2]


@pytest.mark.parametrize("mode", ["dense", "rows"])
def test_gradient_exchange_step_equals_plain_step(mode):
    """The data-parallel step (persistent flat gradient buffer; dense all-reduce or touched-row exchange with the rows
    backward kernel + scatter kernel) at world size 1 trains exactly like the plain eager step."""
    import train_siegel
    common = ["--graph", "grid3d-125", "--manifold", "bounded", "--metric", "riem", "--dims", "3", "--epochs", "8",
              "--batch_size", "512", "--val_every", "2", "--learning_rate", "0.02", "--burnin", "3"]
    _, h_plain = train_siegel.train(train_siegel.parser().parse_args(common + ["--no_graph_step"]), log=lambda *_: None)
    _, h_ex = train_siegel.train(train_siegel.parser().parse_args(common + ["--grad_exchange", mode]), log=lambda *_: None)
    assert len(h_plain) == len(h_ex) == 4
    for a, b in zip(h_ex, h_plain):
        assert abs(a[1] - b[1]) < 1e-8 * abs(b[1]) and abs(a[2] - b[2]) < 1e-8 * abs(b[2]), (a, b)


def test_spd_model_trains():
    """configs[4]'s model family end to end: spd forward kernel, fused backward rows + scatter, geoopt-style RSGD step
    (retr(x, u) = sym(x + u + u x^-1 u / 2)): the distortion of the 125-node grid drops and every point stays SPD."""
    import train_siegel
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", "spd", "--metric", "riem",
                                             "--dims", "3", "--epochs", "30", "--batch_size", "512",
                                             "--val_every", "10", "--learning_rate", "0.02", "--burnin", "5"])
    model, hist = train_siegel.train(args, log=lambda *_: None)
    first, last = hist[0][2], hist[-1][2]
    assert last < 0.7 * first, hist
    ok, point, reason = model.check_all_points()
    assert ok, reason


def test_riemannian_adam_trains():
    """train.py:69-70 `--optim radam`: RiemannianAdam (HIP egrad2rgrad / inner / projx kernels) embeds the grid."""
    import train_siegel
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", "upper", "--metric", "riem",
                                             "--dims", "2", "--epochs", "30", "--batch_size", "512", "--optim", "radam",
                                             "--val_every", "5", "--learning_rate", "0.03", "--burnin", "0"])
    model, hist = train_siegel.train(args, log=lambda *_: None)
    assert hist[-1][2] < 0.75 * hist[0][2] and hist[-1][2] < 0.55, hist       # measured: 0.68 (epoch 5) -> 0.42
    ok, point, reason = model.check_all_points()
    assert ok, reason


@pytest.mark.parametrize("manifold,dims", [("upper", 10), ("bounded", 9), ("spd", 16)])
def test_training_through_the_sixteen_lanes_kernels(manifold, dims):
    """dims 9..16 (Siegel) and spd n = 16 (configs[4]'s matrix size) end to end: the forward, the fused loss + backward with
    in-kernel scatter / rows + scatter and the optimiser step all run on the sixteen-lanes-per-pair kernels; the
    distortion of the 125-node grid drops and every point stays on the manifold."""
    import train_siegel
    lr = "0.01" if manifold == "spd" else "0.02"
    args = train_siegel.parser().parse_args(["--graph", "grid3d-125", "--manifold", manifold, "--metric", "riem",
                                             "--dims", str(dims), "--epochs", "20", "--batch_size", "512",
                                             "--val_every", "10", "--learning_rate", lr, "--burnin", "5"])
    model, hist = train_siegel.train(args, log=lambda *_: None)
    first, last = hist[0][2], hist[-1][2]
    assert last < 0.8 * first, hist
    ok, point, reason = model.check_all_points()
    assert ok, reason
