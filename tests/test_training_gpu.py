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


def test_graphed_step_with_the_split_backward_equals_eager_step(monkeypatch):
    """dims 8, upper model, batches of 2 048 pairs: the replayed step takes the split backward (two kernels, persistent workspace
    held by the step, batches sorted by source row) -- inside the captured graph and in the eager ragged batch -- and trains like
    the same harness on the one-launch kernels (SYMPA_SIEGEL_BWD_NO_WORKSPACE=1): same distortion history to rounding."""
    import train_siegel
    common = ["--graph", "grid3d-125", "--manifold", "upper", "--metric", "riem", "--dims", "8", "--epochs", "6",
              "--batch_size", "2048", "--val_every", "2", "--learning_rate", "0.02", "--burnin", "2"]
    _, h_split = train_siegel.train(train_siegel.parser().parse_args(common), log=lambda *_: None)
    monkeypatch.setenv("SYMPA_SIEGEL_BWD_NO_WORKSPACE", "1")
    _, h_one = train_siegel.train(train_siegel.parser().parse_args(common), log=lambda *_: None)
    assert len(h_split) == len(h_one) == 3
    for a, b in zip(h_split, h_one):
        assert a[0] == b[0]
        assert abs(a[1] - b[1]) < 1e-7 * abs(b[1]) and abs(a[2] - b[2]) < 1e-7 * abs(b[2]), (a, b)
    assert h_split[-1][2] < h_split[0][2]


@pytest.mark.parametrize("mode", ["dense", "rows", "sharded"])
def test_gradient_exchange_step_equals_plain_step(mode):
    """The data-parallel step (persistent flat gradient buffer; dense all-reduce, touched-row exchange with the rows
    backward kernel + scatter kernel, or reduce-scatter + sharded step + all-gather; since round 4 replayed as graphs by
    DistributedTrainStep, ragged last batch eager) at world size 1 trains exactly like the plain eager step."""
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


@pytest.mark.parametrize("manifold", ["upper", "bounded"])
def test_riemannian_adam_step_captured_in_a_graph(manifold):
    """`--optim radam` under GraphedTrainStep (classic mode): the bias corrections come from device words the step itself
    advances, the warm-up steps of the capture leave no trace in the moments, and K replays equal K eager steps."""
    from sympa_amd import data
    from sympa_amd.optim import RiemannianAdam
    from sympa_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda:0")
    nodes, b = 300, 2048
    ma = _toy_model(manifold, "riem", 3, nodes, dev)
    mb = _toy_model(manifold, "riem", 3, nodes, dev)
    oa = RiemannianAdam(ma.parameters(), lr=0.02, eps=1e-7, stabilize=None)
    ob = RiemannianAdam(mb.parameters(), lr=0.02, eps=1e-7, stabilize=None)
    stepper = GraphedTrainStep(ma, oa, b, 50.0, dev, two_kernels=False)
    assert stepper.mode == "classic"
    g = torch.Generator().manual_seed(9)
    for it in range(5):
        ids = data.sample_pairs(nodes, b, it, 1).to(dev)
        gd = (torch.rand(b, generator=g, dtype=torch.float64) * 5 + 1).to(dev)
        stepper(ids[:, :2], gd)
        ob.zero_grad(set_to_none=False)
        mb.fused_loss_backward(ids[:, :2].contiguous(), gd)
        torch.nn.utils.clip_grad_norm_(mb.parameters(), 50.0)
        ob.step()
        ta, tb = ma.embeddings.embeds.detach(), mb.embeddings.embeds.detach()
        assert float((ta - tb).abs().max()) < 1e-11 * float(tb.abs().max()), (manifold, it)
        assert abs(float(ma.scale.detach()) - float(mb.scale.detach())) < 1e-12
    sa, sb = oa.state[ma.embeddings.embeds], ob.state[mb.embeddings.embeds]
    assert abs(float(sa["bias_pows"][0]) - 0.9 ** 5) < 1e-14 and abs(float(sb["bias_pows"][0]) - 0.9 ** 5) < 1e-14
    assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) < 1e-12 * float(sb["exp_avg"].abs().max())


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


def _toy_model(manifold, metric, dims, nodes, dev, train_scale=True, seed=1):
    from sympa_amd import data
    from sympa_amd.model import Model

    class A:
        pass
    A.manifold, A.metric, A.dims, A.num_points = manifold, metric, dims, nodes
    A.scale_coef, A.scale_init, A.train_scale = 2.0, 1.5, train_scale
    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.trained_like_table(nodes, dims, model=manifold, seed=seed)
    return m.to(dev)


@pytest.mark.parametrize("model,n,nodes", [("upper", 4, 700), ("bounded", 3, 1000), ("upper", 6, 300), ("upper", 1, 70)])
def test_fused_optimiser_kernel_equals_the_separate_kernels(model, n, nodes):
    """sympa_rsgd_step_fused (clip norm with a grid barrier + table step + scale / weight step + zero_grad + step counter,
    one launch) == sympa_sqnorm_accum + sympa_rsgd_step_clipped + sympa_sgd_step_clipped + memsets, with and without an
    active clip; the workspace is left ready for the next call; more row blocks than CUs are refused."""
    from sympa_amd import _lib, data, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3 + n)
    table0 = data.trained_like_table(nodes, n, model=model, seed=2).to(dev)
    grad0 = torch.randn(table0.shape, generator=g, dtype=torch.float64)
    grad0 = (0.5 * (grad0 + grad0.transpose(-1, -2))).to(dev)
    scale0 = torch.tensor([1.3], dtype=torch.float64, device=dev)
    gs0 = torch.tensor([0.7], dtype=torch.float64, device=dev)
    w0 = torch.linspace(0.2, 1.1, n, dtype=torch.float64).to(dev)
    gw0 = torch.linspace(-0.5, 0.4, n, dtype=torch.float64).to(dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for max_norm, lr in ((1e9, 1e-3), (0.5, 0.3), (None, 1e-2)):
        # separate kernels
        t1, s1, w1 = table0.clone(), scale0.clone(), w0.clone()
        c1 = torch.zeros(1, dtype=torch.int32, device=dev)
        if max_norm is not None:
            sq = torch.zeros(1, dtype=torch.float64, device=dev)
            for t in (grad0, gs0, gw0):
                ops.sqnorm_accum_(t, sq)
            ops.rsgd_step_(t1, grad0, model, lr, 1e-3, counter=c1, clip_sqnorm=sq, max_norm=max_norm)
            ops.sgd_step_clipped_(s1, gs0, 0.5 * lr, 0.0, clip_sqnorm=sq, max_norm=max_norm)
            ops.sgd_step_clipped_(w1, gw0, lr, 1e-2, clip_sqnorm=sq, max_norm=max_norm)
        else:
            ops.rsgd_step_(t1, grad0, model, lr, 1e-3, counter=c1)
            ops.sgd_step_clipped_(s1, gs0, 0.5 * lr, 0.0)
            ops.sgd_step_clipped_(w1, gw0, lr, 1e-2)
        # one kernel
        t2, s2, w2 = table0.clone(), scale0.clone(), w0.clone()
        g2, gs2, gw2 = grad0.clone(), gs0.clone(), gw0.clone()
        c2 = torch.zeros(1, dtype=torch.int32, device=dev)
        fs = ops.FusedStep(t2, g2, model, [(s2, gs2), (w2, gw2)], counter=counter, projected=c2)
        before = int(counter)
        fs.run(lr, 1e-3, max_norm, [0.5 * lr, lr], [0.0, 1e-2])
        ops.check_status(dev)
        assert int(counter) == before + 1
        scale_t = float(t1.abs().max())
        assert float((t1 - t2).abs().max()) < 1e-11 * scale_t, (max_norm, lr)
        assert abs(float(s1 - s2)) < 1e-13 and float((w1 - w2).abs().max()) < 1e-13
        assert int(c1) == int(c2)
        assert float(g2.abs().max()) == 0.0 and float(gs2.abs().max()) == 0.0 and float(gw2.abs().max()) == 0.0
        assert int(fs.ws.view(torch.int32)[:2].abs().sum()) == 0          # barrier words reset
        if max_norm is not None:
            # the same step with the squared-norm partials handed in (what the deterministic graph does: no pass over the
            # gradient, no grid barrier): identical coefficient up to the order of the sum
            t4, s4, w4 = table0.clone(), scale0.clone(), w0.clone()
            g4, gs4, gw4 = grad0.clone(), gs0.clone(), gw0.clone()
            parts = torch.stack(((grad0 * grad0).sum(), (gs0 * gs0).sum() + (gw0 * gw0).sum(),
                                 torch.zeros((), dtype=torch.float64, device=dev)))
            ops.FusedStep(t4, g4, model, [(s4, gs4), (w4, gw4)], sq_partials=parts).run(lr, 1e-3, max_norm, [0.5 * lr, lr], [0.0, 1e-2])
            assert float((t4 - t1).abs().max()) < 1e-11 * scale_t and abs(float(s4 - s1)) < 1e-13
            assert float(g4.abs().max()) == 0.0
        # a second step with the same object: the workspace needs no host-side reset
        g2.copy_(grad0); gs2.copy_(gs0); gw2.copy_(gw0)
        t3 = t2.clone()
        fs.run(lr, 1e-3, max_norm, [0.5 * lr, lr], [0.0, 1e-2])
        assert torch.isfinite(t2).all() and not torch.equal(t2, t3)
    big = torch.zeros(300 * 256, 2, 1, 1, dtype=torch.float64, device=dev)
    big[:, 1] = 1.0
    assert not ops.FusedStep.supported(big)
    with pytest.raises(_lib.SympaHipError):
        ops.FusedStep(big, torch.zeros_like(big), "upper").run(1e-3, 0.0, 1.0)


@pytest.mark.parametrize("model,n", [("upper", 4), ("bounded", 2), ("upper", 6), ("upper", 8)])
def test_deterministic_gradient_accumulation(model, n):
    """SURVEY 8f-1's alternative to atomics: per-pair gradient rows (coalesced through the LDS tile) + segmented sum in a
    precomputed order + fixed-order scalar sums.  Two runs give bit-identical embeds.grad, loss and scale gradient; they
    equal the fp64-atomic scatter to 1e-12 (relative to the largest entry); a small case equals the sum evaluated
    sequentially in slot order on the host bit for bit; the step-counter window picks the right batch."""
    from sympa_amd import data, ops
    dev = torch.device("cuda:0")
    nodes, b, steps = 211, 4099 if n < 8 else 1500, 3
    g = torch.Generator().manual_seed(40 + n)
    table = data.trained_like_table(nodes, n, model=model, seed=4).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (steps * b,), generator=g), torch.randint(0, nodes, (steps * b,), generator=g)), 1).to(dev)
    gd = torch.randint(1, 9, (steps * b,), generator=g).to(torch.float64).to(dev)
    sc = torch.tensor([1.7], dtype=torch.float64, device=dev)
    w = torch.linspace(0.3, 1.2, n, dtype=torch.float64).to(dev)
    order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0].view(steps, b), trip[:, 1].view(steps, b)), dim=1), nodes)
    assert order.shape == (steps, 2 * b) and rowptr.shape == (steps, nodes + 1) and int(rowptr[:, -1].min()) == 2 * b
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for metric in ("riem", "wsum"):
        for step in (0, 2):
            counter.fill_(step)

            def det():
                rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
                part = torch.empty((b + 63) // 64, 2 + n, dtype=torch.float64, device=dev)
                loss, gs = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
                gw = torch.zeros(n, dtype=torch.float64, device=dev)
                ops.model_train_backward(table, trip, gd, b, loss, model, metric, w, gw, sc, gs, 2.0, 1.0, grad_rows=rows,
                                         step_counter=counter, wave_partials=part)
                grad = torch.full_like(table, 7.0)          # overwritten, not accumulated into
                ops.segment_sum_rows_(grad, rows, order, rowptr, step_counter=counter, wave_partials=part,
                                      num_waves=(b + 63) // 64, partial_stride=2 + n, loss=loss, grad_scale=gs,
                                      grad_weights=gw if metric == "wsum" else None)
                return grad, loss, gs, gw, rows
            a1, a2 = det(), det()
            for x, y in zip(a1[:4], a2[:4]):
                assert torch.equal(x, y)
            # squared-norm partials of the finished gradient (+ the scalar gradients), bitwise reproducible
            sq = [torch.zeros(ops.segment_sum_partials(table), dtype=torch.float64, device=dev) for _ in range(2)]
            for q in sq:
                part = torch.empty((b + 63) // 64, 2 + n, dtype=torch.float64, device=dev)
                loss_, gs_, gw_ = (torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev),
                                   torch.zeros(n, dtype=torch.float64, device=dev))
                ops.model_train_backward(table, trip, gd, b, loss_, model, metric, w, gw_, sc, gs_, 2.0, 1.0, grad_rows=a1[4],
                                         step_counter=counter, wave_partials=part)
                ops.segment_sum_rows_(torch.empty_like(table), a1[4], order, rowptr, step_counter=counter, wave_partials=part,
                                      num_waves=(b + 63) // 64, partial_stride=2 + n, loss=loss_, grad_scale=gs_,
                                      grad_weights=gw_ if metric == "wsum" else None, sq_partials=q)
            assert torch.equal(sq[0], sq[1])
            want_sq = float((a1[0] ** 2).sum() + a1[2] ** 2 + ((a1[3] ** 2).sum() if metric == "wsum" else 0.0))
            assert abs(float(sq[0].sum()) - want_sq) < 1e-12 * want_sq
            # the atomic form of the same batch
            grad = torch.zeros_like(table)
            loss, gs = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
            gw = torch.zeros(n, dtype=torch.float64, device=dev)
            sl = slice(step * b, (step + 1) * b)
            ops.model_loss_backward(table, trip[sl], gd[sl], grad, loss, model, metric, w, gw, sc, gs, 2.0, 1.0)
            ops.check_status(dev)
            big = float(grad.abs().max())
            tol = 1e-12 if n <= 6 else 1e-8      # n = 8: the atomic form runs the eight-lanes-per-pair kernel (another route)
            assert float((a1[0] - grad).abs().max()) < tol * big, (metric, step)
            assert abs(float(a1[1] - loss)) < tol * abs(float(loss)) and abs(float(a1[2] - gs)) < 10 * tol * abs(float(gs))
            if metric == "wsum":
                assert float((a1[3] - gw).abs().max()) < 10 * tol * float(gw.abs().max())
    # bit for bit the sequential sum in slot order (rows of the last det() call, batch 2)
    rows_c = a1[4].cpu().reshape(2 * b, -1)
    o, rp = order[2].cpu(), rowptr[2].cpu()
    want = torch.zeros(nodes, rows_c.shape[1], dtype=torch.float64)
    for r in range(0, nodes, 17):
        s = torch.zeros(rows_c.shape[1], dtype=torch.float64)
        for p in range(int(rp[r]), int(rp[r + 1])):
            s = s + rows_c[int(o[p])]
        want[r] = s
        assert torch.equal(a1[0][r].cpu().reshape(-1), want[r]), r
    # accumulate / alpha
    acc = torch.ones_like(table)
    ops.segment_sum_rows_(acc, a1[4], order[2].contiguous(), rowptr[2].contiguous(), alpha=0.5, accumulate=True)
    assert float((acc - (1.0 + 0.5 * a1[0])).abs().max()) < 1e-13 * max(1.0, float(a1[0].abs().max()))


@pytest.mark.parametrize("n", [2, 4, 6])
@pytest.mark.parametrize("tail", [1, 65, 129, 192])
def test_wave_partials_are_not_written_past_their_last_row(n, tail):
    """The per-wave sums are [ceil(b / 64)][2 + n]; the last 256-thread block has up to three waves with no live pair,
    which must not store anything (round-3 ADVICE: they wrote up to 3 (2 + n) doubles past the end).  Sentinel rows
    after the buffer stay untouched for b % 256 in {1, 65, 129, 192}."""
    from sympa_amd import data, ops
    dev = torch.device("cuda:0")
    nodes, b = 97, 512 + tail
    g = torch.Generator().manual_seed(n * 1000 + tail)
    table = data.trained_like_table(nodes, n, model="upper", seed=5).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (b,), generator=g), torch.randint(0, nodes, (b,), generator=g)), 1).to(dev)
    gd = torch.randint(1, 9, (b,), generator=g).to(torch.float64).to(dev)
    waves = (b + 63) // 64
    buf = torch.full((waves + 4, 2 + n), -777.0, dtype=torch.float64, device=dev)
    rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.model_train_backward(table, trip, gd, b, loss, "upper", "riem", None, None, None, None, 1.0, 1.0, grad_rows=rows,
                             wave_partials=buf[:waves])
    ops.check_status(dev)
    torch.cuda.synchronize()
    assert bool((buf[waves:] == -777.0).all()), buf[waves:]
    assert bool((buf[:waves] != -777.0).all())


@pytest.mark.parametrize("model,n", [("upper", 7), ("upper", 8), ("bounded", 8), ("upper", 11), ("bounded", 9)])
def test_step_counter_window_in_the_lanes_per_pair_backward_kernels(model, n):
    """Round 4: the training graph's batch window (device step counter) also exists in the eight- / sixteen-lanes-per-pair
    backward kernels (dims 7..16), so the replayed multi-GPU step runs the FAST dims-8 kernel: batch c of a loaded epoch
    through sympa_model_train_backward == sympa_model_loss_backward on that slice (same kernel family: to rounding of the
    atomics), in the scatter and in the per-pair-rows form."""
    from sympa_amd import data, ops
    dev = torch.device("cuda:0")
    nodes, b, steps = 97, 300, 3
    g = torch.Generator().manual_seed(70 + n)
    table = data.trained_like_table(nodes, n, model=model, seed=4).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (steps * b,), generator=g), torch.randint(0, nodes, (steps * b,), generator=g)), 1).to(dev)
    gd = torch.randint(1, 9, (steps * b,), generator=g).to(torch.float64).to(dev)
    sc = torch.tensor([1.7], dtype=torch.float64, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for step in (0, 2):
        counter.fill_(step)
        sl = slice(step * b, (step + 1) * b)
        want = torch.zeros_like(table)
        lw, gsw = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
        ops.model_loss_backward(table, trip[sl], gd[sl], want, lw, model, "riem", None, None, sc, gsw, 2.0, 1.0)
        got = torch.zeros_like(table)
        lg, gsg = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
        ops.model_train_backward(table, trip, gd, b, lg, model, "riem", None, None, sc, gsg, 2.0, 1.0, grad_table=got,
                                 step_counter=counter)
        ops.check_status(dev)
        big = float(want.abs().max())
        assert float((got - want).abs().max()) < 1e-11 * big, (model, n, step)
        assert abs(float(lg - lw)) < 1e-11 * abs(float(lw)) and abs(float(gsg - gsw)) < 1e-10 * abs(float(gsw))
        rows = torch.zeros(2 * b, 2, n, n, dtype=torch.float64, device=dev)
        lr_ = torch.zeros(1, dtype=torch.float64, device=dev)
        ops.model_train_backward(table, trip, gd, b, lr_, model, "riem", None, None, sc, None, 2.0, 1.0, grad_rows=rows,
                                 step_counter=counter)
        dense = torch.zeros_like(table)
        dense.index_add_(0, trip[sl, 0], rows[:b])
        dense.index_add_(0, trip[sl, 1], rows[b:])
        assert float((dense - want).abs().max()) < 1e-11 * big, (model, n, step, "rows")


def test_two_kernel_epoch_trains_like_the_classic_graph_and_the_deterministic_form_is_reproducible():
    """The two-kernel step driven by the device step counter (load_epoch + run_steps) == round 2's classic graph called
    batch by batch (tolerance: atomics); the deterministic form run twice gives bit-identical tables, scales and losses."""
    from sympa_amd import ops
    from sympa_amd.optim import RiemannianSGD
    from sympa_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda:0")
    nodes, n, batch, steps = 400, 3, 1024, 7
    g = torch.Generator().manual_seed(9)
    trip = torch.stack((torch.randint(0, nodes, (steps * batch + 300,), generator=g),
                        torch.randint(0, nodes, (steps * batch + 300,), generator=g),
                        torch.randint(1, 9, (steps * batch + 300,), generator=g)), 1).to(dev)

    def run(form):
        m = _toy_model("upper", "riem", n, nodes, dev)
        opt = RiemannianSGD(m.parameters(), lr=5e-3)
        st = GraphedTrainStep(m, opt, batch, 2.0, dev, two_kernels=form != "classic", deterministic=form == "det",
                              accumulate_loss=True)
        assert st.mode == ("classic" if form == "classic" else "two_kernels")
        for epoch in range(2):
            if form == "classic":
                for s in range(0, trip.shape[0], batch):
                    st(trip[s:s + batch, :2], trip[s:s + batch, 2].to(torch.float64))
            else:
                full = st.load_epoch(trip)
                assert full == steps
                st.run_steps()
                st(trip[full * batch:, :2], trip[full * batch:, 2].to(torch.float64))      # ragged remainder
        ops.check_status(dev)
        return m.embeddings.embeds.detach().clone(), m.scale.detach().clone(), st.loss.clone()
    classic, two, det1, det2 = run("classic"), run("two"), run("det"), run("det")
    for a, b_ in zip(det1, det2):
        assert torch.equal(a, b_)
    for other in (two, det1):
        assert float((other[0] - classic[0]).abs().max()) < 1e-9
        assert abs(float(other[1] - classic[1])) < 1e-9 and abs(float(other[2] - classic[2])) < 1e-8 * abs(float(classic[2]))
    assert float((classic[0] - _toy_model("upper", "riem", n, nodes, dev).embeddings.embeds.detach()).abs().max()) > 1e-4


@pytest.mark.parametrize("model,metric", [("upper", "riem"), ("bounded", "wsum")])
def test_two_kernel_radam_epoch_equals_the_optimiser_called_step_by_step(model, metric):
    """`--optim radam` through the two-kernel step (C-ABI sympa_radam_step_fused: clip + RiemannianAdam of the table + Adam of
    the scale / the wsum weights + zero_grad + step counter in one launch, the powers b^t advanced in the kernel) == backward,
    clip_grad_norm_ and RiemannianAdam.step() called batch by batch; the deterministic form is bitwise reproducible."""
    from sympa_amd import ops
    from sympa_amd.optim import RiemannianAdam
    from sympa_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda:0")
    nodes, n, batch, steps = 400, 3, 1024, 6
    g = torch.Generator().manual_seed(19)
    trip = torch.stack((torch.randint(0, nodes, (steps * batch + 200,), generator=g),
                        torch.randint(0, nodes, (steps * batch + 200,), generator=g),
                        torch.randint(1, 9, (steps * batch + 200,), generator=g)), 1).to(dev)

    def run(form):
        m = _toy_model(model, metric, n, nodes, dev)
        opt = RiemannianAdam(m.parameters(), lr=0.01, eps=1e-7, stabilize=None)
        if form == "eager":
            for epoch in range(2):
                for s in range(0, trip.shape[0], batch):
                    opt.zero_grad(set_to_none=False)
                    m.fused_loss_backward(trip[s:s + batch, :2].contiguous(), trip[s:s + batch, 2].to(torch.float64))
                    torch.nn.utils.clip_grad_norm_(m.parameters(), 2.0)
                    opt.step()
        else:
            st = GraphedTrainStep(m, opt, batch, 2.0, dev, deterministic=form == "det", accumulate_loss=True)
            assert st.mode == "two_kernels"
            for epoch in range(2):
                full = st.load_epoch(trip)
                assert full == steps
                st.run_steps()
                st(trip[full * batch:, :2], trip[full * batch:, 2].to(torch.float64))      # ragged remainder
        ops.check_status(dev)
        pows = opt.state[m.embeddings.embeds]["bias_pows"].clone()
        extra = [p.detach().clone() for p in m.parameters() if p is not m.embeddings.embeds]
        return m.embeddings.embeds.detach().clone(), extra, pows
    eager, two, det1, det2 = run("eager"), run("two"), run("det"), run("det")
    assert torch.equal(det1[0], det2[0]) and all(torch.equal(a, b_) for a, b_ in zip(det1[1], det2[1]))
    t = 2 * (steps + 1)
    for other in (two, det1):
        assert float((other[0] - eager[0]).abs().max()) < 1e-8 * float(eager[0].abs().max())
        for a, b_ in zip(other[1], eager[1]):
            assert float((a - b_).abs().max()) < 1e-8 * max(1.0, float(b_.abs().max()))
        assert abs(float(other[2][0]) - 0.9 ** t) < 1e-13 and abs(float(other[2][1]) - 0.999 ** t) < 1e-13
    assert float((eager[0] - _toy_model(model, metric, n, nodes, dev).embeddings.embeds.detach()).abs().max()) > 1e-3


def test_graphed_adam_step_follows_a_restored_optimizer_state_and_an_eager_step_leaves_the_batch_window_alone():
    """Round-3 ADVICE: (i) `opt.load_state_dict` replaces RiemannianAdam's moment / power tensors; the replayed graph and
    the fused kernel's plan hold their old addresses -> GraphedTrainStep must notice and rebuild, so that a resumed run
    equals the uninterrupted one.  (ii) An eager step between `load_epoch` and the end of `run_steps` must not shift the
    device step counter that addresses the loaded batches."""
    import copy
    from sympa_amd import ops
    from sympa_amd.optim import RiemannianAdam
    from sympa_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda:0")
    nodes, n, batch, steps = 300, 3, 512, 6
    g = torch.Generator().manual_seed(23)
    trip = torch.stack((torch.randint(0, nodes, (steps * batch,), generator=g),
                        torch.randint(0, nodes, (steps * batch,), generator=g),
                        torch.randint(1, 9, (steps * batch,), generator=g)), 1).to(dev)
    extra = torch.stack((torch.randint(0, nodes, (100,), generator=g), torch.randint(0, nodes, (100,), generator=g),
                         torch.randint(1, 9, (100,), generator=g)), 1).to(dev)

    def make():
        m = _toy_model("upper", "riem", n, nodes, dev)
        opt = RiemannianAdam(m.parameters(), lr=0.01, eps=1e-7, stabilize=None)
        return m, opt, GraphedTrainStep(m, opt, batch, 2.0, dev, deterministic=True)
    # uninterrupted: 3 steps, an eager ragged batch in the middle of the window, 3 more steps
    m1, o1, s1 = make()
    assert s1.mode == "two_kernels"
    s1.load_epoch(trip)
    s1.run_steps(3)
    s1(extra[:, :2], extra[:, 2].to(torch.float64))
    s1.run_steps(3)
    assert int(s1.counter) == steps
    # the same with a checkpoint after the eager step: model + optimiser state into fresh objects, then the last 3 batches
    m2, o2, s2 = make()
    s2.load_epoch(trip)
    s2.run_steps(3)
    s2(extra[:, :2], extra[:, 2].to(torch.float64))
    sd_m, sd_o = copy.deepcopy(m2.state_dict()), copy.deepcopy(o2.state_dict())
    m3, o3, s3 = make()
    s3.load_epoch(trip[:batch])
    s3.run_steps(1)                     # a captured graph and a plan exist, holding the addresses of o3's first state
    m3.load_state_dict(sd_m)
    o3.load_state_dict(sd_o)            # replaces the state tensors
    s3.load_epoch(trip[3 * batch:])
    s3.run_steps(3)
    ops.check_status(dev)
    a, b_ = m1.embeddings.embeds.detach(), m3.embeddings.embeds.detach()
    assert float((a - b_).abs().max()) < 1e-12 * float(a.abs().max())
    t1, t3 = o1.state[m1.embeddings.embeds], o3.state[m3.embeddings.embeds]
    assert float((t1["bias_pows"] - t3["bias_pows"]).abs().max()) < 1e-15
    assert float((t1["exp_avg"] - t3["exp_avg"]).abs().max()) < 1e-12 * max(1e-30, float(t1["exp_avg"].abs().max()))


def test_harness_deterministic_training_is_bitwise_reproducible():
    import train_siegel
    common = ["--graph", "grid3d-125", "--manifold", "bounded", "--metric", "fone", "--dims", "2", "--epochs", "6",
              "--batch_size", "512", "--val_every", "2", "--learning_rate", "0.02", "--burnin", "2", "--train_scale",
              "--deterministic"]
    m1, h1 = train_siegel.train(train_siegel.parser().parse_args(common), log=lambda *_: None)
    m2, h2 = train_siegel.train(train_siegel.parser().parse_args(common), log=lambda *_: None)
    assert h1 == h2
    assert torch.equal(m1.embeddings.embeds, m2.embeddings.embeds) and torch.equal(m1.scale, m2.scale)
    assert h1[-1][2] < h1[0][2]


@pytest.mark.parametrize("n", [7, 8, 10])
def test_classic_graph_addresses_its_batches_through_the_step_counter(n):
    """Round 6: the classic graph of the Siegel models (dims >= 7: split backward at 7, 8, sixteen lanes at 10) reads its batch
    through the device step counter like the two-kernel step, so `load_epoch` + `run_steps` work there too and no batch is copied
    in per step.  Same tables, scale and loss as the same object called batch by batch (tolerance: atomics), with an eager ragged
    batch at the end of each epoch and a per-batch call interleaved (it must not shift the window)."""
    from sympa_amd import ops
    from sympa_amd.optim import RiemannianSGD
    from sympa_amd.train_step import GraphedTrainStep
    dev = torch.device("cuda:0")
    nodes, batch, steps = 300, 2048, 3
    g = torch.Generator().manual_seed(19)
    total = steps * batch + 200
    trip = torch.stack((torch.randint(0, nodes, (total,), generator=g), torch.randint(0, nodes, (total,), generator=g),
                        torch.randint(1, 9, (total,), generator=g)), 1).to(dev)

    def run(form):
        m = _toy_model("upper", "riem", n, nodes, dev)
        opt = RiemannianSGD(m.parameters(), lr=5e-3)
        st = GraphedTrainStep(m, opt, batch, 2.0, dev, accumulate_loss=True)
        assert st.mode == "classic" and st._classic_windowed()
        for epoch in range(2):
            if form == "calls":
                for s in range(0, total, batch):
                    st(trip[s:s + batch, :2], trip[s:s + batch, 2].to(torch.float64))
            else:
                full = st.load_epoch(trip)
                assert full == steps
                st.run_steps(2)
                st.run_steps()
                st(trip[full * batch:, :2], trip[full * batch:, 2].to(torch.float64))      # ragged remainder: eager
        ops.check_status(dev)
        return m.embeddings.embeds.detach().clone(), m.scale.detach().clone(), st.loss.clone()

    a, b_ = run("calls"), run("epoch")
    assert float((a[0] - b_[0]).abs().max()) < 1e-9
    assert abs(float(a[1] - b_[1])) < 1e-9 and abs(float(a[2] - b_[2])) < 1e-8 * abs(float(a[2]))
    assert float((a[0] - _toy_model("upper", "riem", n, nodes, dev).embeddings.embeds.detach()).abs().max()) > 1e-4
