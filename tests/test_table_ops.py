"""Optimiser-side manifold operations: egrad2rgrad, projx, fused RiemannianSGD step.
CPU: the per-row arithmetic (g++ build) against the reference goldens and the oracle.
GPU: the HIP kernels through the C-ABI against the same, plus the optimiser class on a Model."""
import numpy as np
import pytest
import torch

from oracle import siegel_oracle as so
from tests.helpers import GOLDEN, MODELS, T, hostsim_table, points


def relmax(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])
def test_hostsim_against_reference_goldens(n):
    g = np.load(f"{GOLDEN}/primitives_n{n}.npz")
    assert relmax(hostsim_table("egrad2rgrad", "upper", g["upper_pts"], g["grad_in"])[0], g["upper_egrad2rgrad"]) < 1e-13
    assert relmax(hostsim_table("egrad2rgrad", "bounded", g["cayley_upper"], g["grad_in"])[0],
                  g["bounded_egrad2rgrad"]) < 1e-13
    out, moved = hostsim_table("projx", "upper", g["projx_in"])
    assert relmax(out, g["upper_projx"]) < 1e-12
    out, moved = hostsim_table("projx", "bounded", g["bounded_projx_in"])
    assert relmax(out, g["bounded_projx"]) < 1e-9


def step_inputs(model, n, g):
    table = points(model, 50, n, 0.3, g)
    grad = torch.randn(50, 2, n, n, generator=g, dtype=torch.float64)
    grad = 0.5 * (grad + grad.transpose(-1, -2))
    return table, grad


@pytest.mark.parametrize("n", [2, 3, 4])
@pytest.mark.parametrize("model", MODELS)
def test_hostsim_rsgd_step_matches_oracle(model, n):
    g = torch.Generator().manual_seed(3 + n)
    table, grad = step_inputs(model, n, g)
    for lr in (1e-2, 0.7):       # the large step pushes points off the manifold -> projection path
        want, keep = so.rsgd_step(model, table, grad, lr, 0.01)
        out, moved = hostsim_table("rsgd", model, table.numpy(), grad.numpy(), lr=lr, wd=0.01)
        assert relmax(out, want.numpy()) < 1e-9, (model, n, lr)
        assert moved == int((~keep).sum())
    # already-inside points are left bit-identical apart from the symmetrisation (upper_half.py:64)
    out, moved = hostsim_table("projx", model, table.numpy())
    assert moved == 0 and np.array_equal(out, so.to_symmetric(table).numpy())


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])     # 5, 6, 7: round 6 (tools/make_golden.py --round6)
def test_gpu_manifold_methods_against_goldens(dev, n):
    from sympa_amd.manifolds import BoundedDomainManifold, UpperHalfManifold
    g = np.load(f"{GOLDEN}/primitives_n{n}.npz")
    up, bd = UpperHalfManifold(dims=n), BoundedDomainManifold(dims=n)
    u = T(g["grad_in"]).to(dev)
    assert relmax(up.egrad2rgrad(T(g["upper_pts"]).to(dev), u).cpu(), g["upper_egrad2rgrad"]) < 1e-13
    assert relmax(bd.egrad2rgrad(T(g["cayley_upper"]).to(dev), u).cpu(), g["bounded_egrad2rgrad"]) < 1e-13
    assert relmax(up.projx(T(g["projx_in"]).to(dev)).cpu(), g["upper_projx"]) < 1e-12
    assert up.projected_points == int((~torch.from_numpy(g["pcp_keep"])).numel()) or up.projected_points > 0
    assert relmax(bd.projx(T(g["bounded_projx_in"]).to(dev)).cpu(), g["bounded_projx"]) < 1e-9
    # retr(x, u) = projx(x + u)   (siegel_manifold.py:74-87)
    x = T(g["upper_pts"]).to(dev)
    assert torch.equal(up.retr(x, 0.01 * u), up.projx(x + 0.01 * u))


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_gpu_rsgd_optimizer_on_model(dev, model):
    """geoopt-style RiemannianSGD over a Model: table rows by the fused kernel, scale by Euclidean SGD;
    two steps of the reference's training loop against the oracle composition on the CPU."""
    from sympa_amd import ops
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianSGD

    class A:
        manifold, metric, dims, num_points = model, "riem", 3, 60
        scale_coef, scale_init, train_scale = 1.0, 1.0, True

    g = torch.Generator().manual_seed(21)
    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = points(model, 60, 3, 0.3, g)
    m = m.to(dev)
    opt = RiemannianSGD(m.parameters(), lr=0.05, weight_decay=0.0, stabilize=None)
    table = m.embeddings.embeds.detach().cpu().clone()
    scale = m.scale.detach().cpu().clone()
    for it in range(2):
        trip = torch.stack((torch.randint(0, 60, (400,), generator=g), torch.randint(0, 60, (400,), generator=g)), 1)
        trip = trip[trip[:, 0] != trip[:, 1]]
        gd = torch.randint(1, 7, (trip.shape[0],), generator=g).to(torch.float64)
        opt.zero_grad()
        loss = m.fused_loss_backward(trip.to(dev), gd.to(dev))
        opt.step()
        # oracle: autograd + restated optimiser on the CPU
        t = table.clone().requires_grad_(True)
        s = scale.clone().requires_grad_(True)
        ref = so.distortion_loss(gd, so.model_forward(t, trip, model, "riem", scale=s, scale_coef=1.0))
        ref.backward()
        gsym = 0.5 * (t.grad + t.grad.transpose(-1, -2))
        table = so.rsgd_step(model, table, gsym, 0.05)[0]
        scale = scale - 0.05 * s.grad
        assert abs(float(loss) - float(ref)) < 1e-8 * abs(float(ref))
        assert relmax(m.embeddings.embeds.detach().cpu(), table) < 1e-7
        assert relmax(m.scale.detach().cpu(), scale) < 1e-8
    ops.check_status(dev)
    assert isinstance(m.manifold.projected_points, int)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("max_norm", [0.05, 1e6])
def test_gpu_clip_folded_into_the_step_equals_torch_clip(dev, model, max_norm):
    """RiemannianSGD.clip_max_norm (the total norm accumulated on the device, the factor applied inside the update
    kernels) == torch.nn.utils.clip_grad_norm_ followed by the plain step (runner.py:115-116), with the clip active
    (max_norm = 0.05) and inactive (1e6); table, scale and wsum weights are all trained."""
    import copy
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianSGD

    class A:
        manifold, metric, dims, num_points = model, "wsum", 3, 50
        scale_coef, scale_init, train_scale = 1.0, 1.3, True

    g = torch.Generator().manual_seed(33)
    m1 = Model(A)
    with torch.no_grad():
        m1.embeddings.embeds.data = points(model, 50, 3, 0.3, g)
    m1 = m1.to(dev)
    m2 = copy.deepcopy(m1)
    trip = torch.stack((torch.randint(0, 50, (300,), generator=g), torch.randint(0, 50, (300,), generator=g)), 1)
    trip = trip[trip[:, 0] != trip[:, 1]].to(dev)
    gd = torch.randint(1, 7, (trip.shape[0],), generator=g).to(torch.float64).to(dev)
    o1 = RiemannianSGD(m1.parameters(), lr=0.05)
    o2 = RiemannianSGD(m2.parameters(), lr=0.05)
    for _ in range(2):
        o1.zero_grad(); o2.zero_grad()
        m1.fused_loss_backward(trip, gd)
        m2.fused_loss_backward(trip, gd)
        total = torch.nn.utils.clip_grad_norm_(m1.parameters(), max_norm)
        o1.step()
        o2.clip_max_norm = max_norm
        o2.step()
        assert (float(total) > max_norm) == (max_norm < 1.0)
        for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
            assert relmax(b.cpu(), a.cpu()) < 1e-10, k


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 7, 8, 11, 16])      # 7, 8: eight lanes per row; 11, 16: sixteen
@pytest.mark.parametrize("model", MODELS)
def test_gpu_tangent_norm_and_riemannian_adam(dev, model, n):
    """manifold.inner(z, u, u) kernel == the restated upper_half.py:68-91 / bounded_domain.py:86-116, and three steps of
    RiemannianAdam (train.py:69-70, --optim radam) == the restated geoopt step on the CPU."""
    from sympa_amd import ops
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianAdam
    g = torch.Generator().manual_seed(40 + n)
    z = points(model, 80, n, 0.3, g)
    u = torch.randn(80, 2, n, n, generator=g, dtype=torch.float64)       # NOT symmetric: A G A of the bounded model is not
    inner = so.upper_inner if model == "upper" else so.bounded_inner
    assert relmax(ops.tangent_sqnorm(z.to(dev), u.to(dev), model).cpu(), inner(z, u)) < 1e-11
    ops.check_status(dev)

    class A:
        manifold, metric, dims, num_points = model, "riem", n, 50
        scale_coef, scale_init, train_scale = 1.0, 1.0, True

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = points(model, 50, n, 0.3, g)
    m = m.to(dev)
    opt = RiemannianAdam(m.parameters(), lr=0.01, eps=1e-7, stabilize=None)
    table = m.embeddings.embeds.detach().cpu().clone()
    scale = m.scale.detach().cpu().clone()
    st = {"step": 0, "exp_avg": torch.zeros_like(table), "exp_avg_sq": torch.zeros(50, dtype=torch.float64)}
    ms, vs = torch.zeros_like(scale), torch.zeros_like(scale)
    for it in range(3):
        trip = torch.stack((torch.randint(0, 50, (300,), generator=g), torch.randint(0, 50, (300,), generator=g)), 1)
        trip = trip[trip[:, 0] != trip[:, 1]]
        gd = torch.randint(1, 7, (trip.shape[0],), generator=g).to(torch.float64)
        opt.zero_grad()
        m.fused_loss_backward(trip.to(dev), gd.to(dev))
        opt.step()
        t = table.clone().requires_grad_(True)
        s = scale.clone().requires_grad_(True)
        so.distortion_loss(gd, so.model_forward(t, trip, model, "riem", scale=s, scale_coef=1.0)).backward()
        gsym = 0.5 * (t.grad + t.grad.transpose(-1, -2))
        table = so.radam_step(model, table, gsym, st, 0.01, eps=1e-7)
        ms = 0.9 * ms + 0.1 * s.grad
        vs = 0.999 * vs + 0.001 * s.grad * s.grad
        scale = scale - 0.01 * (ms / (1 - 0.9 ** (it + 1))) / ((vs / (1 - 0.999 ** (it + 1))).sqrt() + 1e-7)
        assert relmax(m.embeddings.embeds.detach().cpu(), table) < 1e-6, (model, n, it)
        assert relmax(m.scale.detach().cpu(), scale) < 1e-7
    ops.check_status(dev)
    ok, _, reason = m.check_all_points()
    assert ok, reason


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 5, 6])
@pytest.mark.parametrize("model", MODELS)
def test_gpu_fused_radam_row_kernel_against_the_separate_kernels(dev, model, n):
    """C-ABI sympa_radam_step (one launch per table, dims <= 6) == egrad2rgrad, moment updates, inner, projx as separate
    kernels / torch ops (what RiemannianAdam runs for dims >= 7), three steps, with weight decay and a step large enough to
    send rows through the projection."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(60 + n)
    rows = 700
    x0 = points(model, rows, n, 0.3, g).to(dev)
    xa, xb = x0.clone(), x0.clone()
    ma, mb = torch.zeros_like(x0), torch.zeros_like(x0)
    va, vb = torch.zeros(rows, dtype=torch.float64, device=dev), torch.zeros(rows, dtype=torch.float64, device=dev)
    pows = torch.ones(2, dtype=torch.float64, device=dev)
    b1, b2, eps, wd = 0.9, 0.999, 1e-7, 0.01
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    for it, lr in enumerate((0.01, 0.3, 0.05)):
        grad = torch.randn(rows, 2, n, n, generator=g, dtype=torch.float64)
        grad = (0.5 * (grad + grad.transpose(-1, -2))).to(dev)
        pows.mul_(torch.tensor([b1, b2], dtype=torch.float64, device=dev))
        ops.radam_step_(xa, grad, ma, va, pows, model, lr, (b1, b2), eps, wd, counter=counter)
        r = ops.egrad2rgrad(xb, grad.add(xb, alpha=wd), model)
        mb.mul_(b1).add_(r, alpha=1 - b1)
        vb.mul_(b2).add_(ops.tangent_sqnorm(xb, r, model), alpha=1 - b2)
        direction = (mb / (1 - b1 ** (it + 1))) / ((vb / (1 - b2 ** (it + 1))).sqrt() + eps).view(-1, 1, 1, 1)
        xb = ops.projx(xb.add(direction, alpha=-lr), model)
        assert relmax(xa.cpu(), xb.cpu()) < 1e-9, (model, n, it)
        assert relmax(ma.cpu(), mb.cpu()) < 1e-11 and relmax(va.cpu(), vb.cpu()) < 1e-11
    ops.check_status(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("n", list(range(7, 17)))      # every instance of the eight- (7, 8) and sixteen-lanes (9..16) row kernels
@pytest.mark.parametrize("model", MODELS)
def test_gpu_table_operations_dims_9_to_16(dev, model, n):
    """dims 9..16: the same row arithmetic as dims <= 8 compiled with rolled loops (siegel_table_rolled.hip) -- egrad2rgrad,
    the RSGD step with a small and a large learning rate (projection path), projx of points already inside (bit-identical
    apart from the symmetrisation) -- against the oracle's restatement of the reference methods."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(40 + n)
    table, grad = step_inputs(model, n, g)
    want = (so.upper_egrad2rgrad if model == "upper" else so.bounded_egrad2rgrad)(table, grad)
    assert relmax(ops.egrad2rgrad(table.to(dev), grad.to(dev), model).cpu(), want) < 1e-12
    for lr in (1e-2, 0.7):
        want, keep = so.rsgd_step(model, table, grad, lr, 0.01)
        tab = table.clone().to(dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.rsgd_step_(tab, grad.to(dev), model, lr, 0.01, counter=cnt)
        ops.check_status(dev)
        assert relmax(tab.cpu(), want) < 1e-8, (model, n, lr)
        assert int(cnt) == int((~keep).sum())
    out = ops.projx(table.to(dev), model)
    assert torch.equal(out.cpu(), so.to_symmetric(table))
    # projx on its own with rows outside the manifold (the gated exact projection) and a non-symmetric input
    bad = table.clone()
    bad[3] = table[3] + 0.01 * torch.randn(2, n, n, generator=g, dtype=torch.float64)
    if model == "upper":
        bad[7, 1] = -bad[7, 1]
        bad[20, 1] = bad[20, 1] - 1.5 * torch.eye(n, dtype=torch.float64)
    else:
        bad[7] = 3.0 * bad[7]
        bad[20] = 1.4 * bad[20] / torch.linalg.matrix_norm(torch.complex(bad[20, 0], bad[20, 1]), ord=2)
    want = (so.upper_projx if model == "upper" else so.bounded_projx)(bad)
    want = want[0] if isinstance(want, tuple) else want
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    got = ops.projx(bad.to(dev), model, counter=cnt)
    ops.check_status(dev)
    assert relmax(got.cpu(), want) < 1e-8 and int(cnt) == 2
    # squared tangent norm (inner) against the restated reference formula
    u = torch.randn(50, 2, n, n, generator=g, dtype=torch.float64)
    inner = so.upper_inner if model == "upper" else so.bounded_inner
    assert relmax(ops.tangent_sqnorm(table.to(dev), u.to(dev), model).cpu(), inner(table, u)) < 1e-10
    ops.check_status(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_gpu_rsgd_step_is_reentrant_across_many_streams(dev, model):
    """Round-3 review: the word the exact projection of the dims >= 7 step is gated on was one of 64 process-global slots
    keyed by stream handle (a mutex, never released, the 65th stream raced).  It is the caller's now (C-ABI `outside_word`).
    200 steps at n = 8 on 100 short-lived streams, all in flight together, every one with rows that leave the interior:
    each table equals the step run alone and every projection is counted; with outside_word = NULL the entry runs the exact
    one-row-per-lane kernel and gives the same table."""
    from sympa_amd import _lib, ops
    n, rows = 8, 50
    g = torch.Generator().manual_seed(77)
    table, grad = step_inputs(model, n, g)
    lr = 0.7                               # pushes rows off the manifold -> the gated projection must run
    want, keep = so.rsgd_step(model, table, grad, lr, 0.01)
    moved = int((~keep).sum())
    assert moved > 0
    gd = grad.to(dev)
    tabs = [table.clone().to(dev) for _ in range(200)]
    cnts = torch.zeros(200, 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    streams = []
    for k, tab in enumerate(tabs):
        if k % 2 == 0:
            streams.append(torch.cuda.Stream())            # > 64 distinct streams alive at once
        with torch.cuda.stream(streams[-1]):
            ops.rsgd_step_(tab, gd, model, lr, 0.01, counter=cnts[k])
    torch.cuda.synchronize()
    ops.check_status(dev)
    assert len({s.cuda_stream for s in streams}) > 1
    for k, tab in enumerate(tabs):
        assert relmax(tab.cpu(), want) < 1e-8, k
    assert bool((cnts == moved).all()), cnts.flatten().tolist()
    # no scratch word: the exact kernel on every row
    lib = _lib.load()
    tab = table.clone().to(dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.zeros(2, dtype=torch.int32, device=dev)
    rc = lib.sympa_rsgd_step(tab.data_ptr(), gd.data_ptr(), rows, n, ops.MODEL_IDS[model], lr, 0.01, 1e-5, cnt.data_ptr(),
                             st.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert relmax(tab.cpu(), want) < 1e-8 and int(cnt) == moved


@pytest.mark.gpu
def test_gpu_sgd_step_of_a_parameter_without_manifold(dev):
    """sympa_sgd_step_clipped (the scale's step inside RiemannianSGD): p <- p - lr (coef g + wd p), coef from the device-side
    squared total norm like clip_grad_norm_."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(1000, generator=g, dtype=torch.float64)
    gr = torch.randn(1000, generator=g, dtype=torch.float64)
    for max_norm in (None, 0.5, 1e6):
        p = p0.clone().to(dev)
        sq = (gr * gr).sum().reshape(1).to(dev) if max_norm is not None else None
        ops.sgd_step_clipped_(p, gr.to(dev), 0.03, 0.01, clip_sqnorm=sq, max_norm=max_norm)
        coef = 1.0 if max_norm is None else min(1.0, max_norm / (float((gr * gr).sum().sqrt()) + 1e-6))
        want = p0 - 0.03 * (coef * gr + 0.01 * p0)
        assert relmax(p.cpu(), want) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("count", [1, 7, 1000, 100003, 5824000])
def test_gpu_sqnorm_accum_sizes_and_alignment(count):
    """sympa_sqnorm_accum (the gradient norm of clip_grad_norm_, runner.py:108): 16-byte loads four at a time with a scalar tail,
    one atomic per block; an 8-byte aligned (not 16) buffer takes the scalar route; the result ACCUMULATES."""
    from sympa_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(count)
    x = torch.randn(count + 1, generator=g, dtype=torch.float64).to(dev)
    for view in (x[:count], x[1:]):
        acc = torch.full((1,), 2.5, dtype=torch.float64, device=dev)
        ops.sqnorm_accum_(view, acc)
        want = 2.5 + float((view.cpu() ** 2).sum())
        assert abs(float(acc) - want) < 1e-11 * want
