"""GPU (`-m gpu`): parity of the HIP path, called through the C-ABI (sympa_amd.ops -> libsympa_hip.so),
against (1) golden vectors produced by the imported reference, (2) the oracle on seeded inputs,
(3) size-independent properties at BASELINE.json's full batch sizes.
Tolerance: 1e-9 relative for distances (north_star: 1e-4), abs floor 1e-12; index handling bit-exact."""
import numpy as np
import pytest
import torch

from oracle import siegel_oracle as so
from tests.helpers import GOLDEN, METRICS, MODELS, T, hostsim_dist, points, rel_err, sym, upper_points

pytestmark = pytest.mark.gpu
TOL = 1e-9
TOL_FAR_VS_REFERENCE = 1e-6   # see tests/test_hostsim_parity.py


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from sympa_amd import _lib
    _lib.load()          # the HIP library must be the thing that runs
    return torch.device("cuda:0")


def gpu_dist(z1, z2, model, metric, w=None, dev="cuda:0", vvd=False):
    from sympa_amd import ops
    r = ops.siegel_dist_forward(T(z1).to(dev), T(z2).to(dev), model, metric,
                                None if w is None else T(w).to(dev), return_vvd=vvd)
    ops.check_status(torch.device(dev))
    return (r[0].cpu(), r[1].cpu()) if vvd else r.cpu()


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 12, 16])     # every kernel family: Jacobi, QL in registers, sixteen lanes
@pytest.mark.parametrize("model", MODELS)
def test_golden_vectors_of_the_reference(dev, model, n):
    g = np.load(f"{GOLDEN}/dist_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        for metric in METRICS:
            got = gpu_dist(z1, z2, model, metric, g["wsum_weights"])
            tol = TOL_FAR_VS_REFERENCE if case in ("far", "s1.0") else TOL
            # d(x, x): exactly 0 here, ~1e-15 per component in the reference (up to 2(n-1) n of them under fmin)
            atol = 1e-10 if (case == "same" and n > 8) else 1e-12
            assert rel_err(got, g[f"{case}__{metric}"], atol=atol) < tol, (model, n, case, metric)
        if f"{case}__vvd_exact50" in g:
            _, vvd = gpu_dist(z1, z2, model, "riem", vvd=True)
            assert rel_err(vvd, g[f"{case}__vvd_exact50"]) < (1e-12 if model == "upper" else 1e-9)


@pytest.mark.parametrize("n", [5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_far_golden_vectors_of_the_reference(dev, model, n):
    """Round 6: the reference's clamp-regime (`far`) outputs at dims 5..8 (tools/make_golden.py --round6) through the dense
    kernels AND the packed-table path, with the 50-digit evaluation of the reference formula for the vector-valued distance."""
    from sympa_amd import ops
    g = np.load(f"{GOLDEN}/dist_far_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        b = z1.shape[0]
        table = torch.cat((T(z1), T(z2))).to(dev)
        ids = torch.stack((torch.arange(b), torch.arange(b) + b), 1).to(dev)
        pack = ops.PackedTable(model)
        for metric in METRICS:
            # 1e-4 = north_star; it is the REFERENCE's fp64 error here (its riem against the 50-digit evaluation of its own formula
            # reaches 8.8e-5 at n = 7, 8), see tests/test_hostsim_parity.py::test_golden_far
            got = gpu_dist(z1, z2, model, metric, g["wsum_weights"])
            assert rel_err(got, g[f"{case}__{metric}"]) < 1e-4, (model, n, case, metric)
            w = T(g["wsum_weights"]).reshape(-1).to(dev)
            packed = ops.model_forward_packed(pack.ensure(table), ids, metric, w)
            ops.check_status(dev)
            assert rel_err(packed.cpu(), g[f"{case}__{metric}"]) < 1e-4, (model, n, case, metric, "packed")
            if metric == "riem":
                exact = np.sqrt((g[f"{case}__vvd_exact50"] ** 2).sum(1))
                assert rel_err(got, exact) < 1e-9 and rel_err(packed.cpu(), exact) < 1e-8, (model, n, case)
        _, vvd = gpu_dist(z1, z2, model, "riem", vvd=True)
        assert rel_err(vvd, g[f"{case}__vvd_exact50"]) < 1e-8, (model, n, case)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_against_oracle_seeded(dev, model, n):
    g = torch.Generator().manual_seed(1000 + n)
    b = 777   # ragged: not a multiple of the 256-lane block
    for s in (1e-3, 0.3, 0.8):
        z1, z2 = points(model, b, n, s, g), points(model, b, n, s, g)
        for metric in METRICS:
            w = torch.linspace(-0.3, 1.2, n)
            got = gpu_dist(z1, z2, model, metric, w)
            want = so.manifold_dist(model, z1, z2, metric, w)
            assert rel_err(got, want) < TOL, (model, n, s, metric)


@pytest.mark.parametrize("model", MODELS)
def test_gpu_matches_cpu_build_of_same_arithmetic(dev, model):
    """hipcc device code vs g++ host build of siegel_math.hpp: same algorithm, agreement to rounding."""
    g = torch.Generator().manual_seed(5)
    z1, z2 = points(model, 300, 4, 0.5, g), points(model, 300, 4, 0.5, g)
    got, vv = gpu_dist(z1, z2, model, "riem", vvd=True)
    ref, vref, _ = hostsim_dist(z1.numpy(), z2.numpy(), model, "riem")
    assert rel_err(got, ref) < 1e-12
    assert rel_err(vv, vref, atol=1e-13) < 1e-10


def test_model_forward_golden_and_bit_exact_gather(dev):
    from sympa_amd import ops
    g = np.load(f"{GOLDEN}/model_forward.npz")
    trip = torch.from_numpy(g["triplets"]).to(dev)          # int64 [b,3]: strided src/dst, no copy
    for model in MODELS:
        table = T(g[f"table_{model}"]).to(dev)
        for scale, coef in ((1.0, 1.0), (0.05, 1.0), (3.0, 2.0)):
            sc = torch.tensor([scale], device=dev)
            got = ops.model_forward(table, trip, model, "riem", None, sc, coef)
            ops.check_status(dev)
            assert rel_err(got.cpu(), g[f"{model}__scale{scale}_coef{coef}"]) < TOL
        # fused in-kernel gather == explicit gather + dist, bit for bit (index work is exact)
        fused = ops.model_forward(table, trip, model, "riem")
        two_step = ops.siegel_dist_forward(table[trip[:, 0]], table[trip[:, 1]], model, "riem")
        assert torch.equal(fused, two_step)
        # [b,2] triplets (runner.py:149 builds those) give the same bits as [b,3]
        assert torch.equal(ops.model_forward(table, trip[:, :2].contiguous(), model, "riem"), fused)


def test_model_class_is_a_drop_in(dev):
    from sympa_amd import ops
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "upper", "wsum", 3, 40
        scale_coef, scale_init, train_scale = 2.0, 1.5, False

    m = Model(A).to(dev)
    # same keys as the reference's Model (the manifold is registered under Model and under Embeddings)
    assert set(m.state_dict().keys()) == {"scale", "embeddings.embeds", "manifold.metric.weights",
                                          "embeddings.manifold.metric.weights"}
    trip = torch.randint(0, 40, (100, 3), device=dev)
    with torch.no_grad():
        out = m(trip)
    ops.check_status(dev)
    want = so.model_forward(m.embeddings.embeds.detach().cpu(), trip.cpu(), "upper", "wsum",
                            m.manifold.metric.weights.detach().cpu(), m.scale.detach().cpu(), A.scale_coef)
    assert rel_err(out.cpu(), want) < TOL
    z1 = m.embeddings(trip[:, 0]).detach()
    z2 = m.embeddings(trip[:, 1]).detach()
    assert rel_err((m.distance(z1, z2) * m.get_scale()).detach().cpu(), want) < TOL


def test_model_forward_follows_replaced_parameters_and_modules(dev):
    """Round-4 advice: forward() resolves its attribute chains once; REPLACING a Parameter object or the metric must not leave it
    on the stale tensors while forward_batches / distortion read the live attributes."""
    from sympa_amd import data, ops
    from sympa_amd.manifolds.metrics import Metric, MetricType
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "upper", "riem", 3, 40
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A).to(dev)
    trip = torch.randint(0, 40, (100, 3), device=dev)
    with torch.no_grad():
        first = m(trip).cpu()
        # a NEW Parameter object (not `.data = ...`): the cached tuple must be rebuilt
        new_table = data.trained_like_table(40, 3, model="upper", seed=5).to(dev)
        m.embeddings.embeds = type(m.embeddings.embeds)(new_table, manifold=m.manifold)
        got = m(trip).cpu()
        want = so.model_forward(new_table.cpu(), trip.cpu(), "upper", "riem")
        assert rel_err(got, want) < TOL and not torch.allclose(got, first)
        assert torch.equal(m.forward_batches([trip])[0].cpu(), got)
        # a new scale Parameter and another metric
        m.scale = torch.nn.Parameter(torch.tensor([3.0], dtype=torch.float64, device=dev), requires_grad=False)
        m.manifold.metric = Metric.get(MetricType.FINSLER_ONE, 3)
        got = m(trip).cpu()
        want = so.model_forward(new_table.cpu(), trip.cpu(), "upper", "fone", None, torch.tensor([3.0], dtype=torch.float64), 1.0)
        assert rel_err(got, want) < TOL
    ops.check_status(dev)


def test_edge_cases(dev):
    from sympa_amd import ops
    g = torch.Generator().manual_seed(9)
    table = upper_points(10, 4, 0.3, g).to(dev)
    # empty batch
    assert ops.model_forward(table, torch.zeros(0, 2, dtype=torch.int64, device=dev)).shape == (0,)
    assert ops.siegel_dist_forward(table[:0], table[:0]).shape == (0,)
    # single pair
    one = ops.model_forward(table, torch.tensor([[1, 2]], device=dev))
    assert one.shape == (1,) and torch.isfinite(one).all()
    ops.check_status(dev)
    # index out of range -> status + NaN, like the reference's IndexError
    bad = ops.model_forward(table, torch.tensor([[1, 2], [3, 10], [-1, 0]], device=dev))
    assert torch.isfinite(bad[0]) and torch.isnan(bad[1]) and torch.isnan(bad[2])
    with pytest.raises(IndexError):
        ops.check_status(dev)
    # point outside the manifold (Im z not PD) -> AssertionError like siegel_manifold.py:64-66
    off = table.clone()
    off[4, 1] = -off[4, 1]
    ops.model_forward(off, torch.tensor([[4, 5], [1, 2]], device=dev))
    with pytest.raises(AssertionError):
        ops.check_status(dev)
    # d(x, x) == 0 exactly (reference test_upper_half.py:128-133 asserts allclose to 0)
    assert torch.all(ops.siegel_dist_forward(table, table) == 0)
    # unsupported dims fail loudly
    with pytest.raises(RuntimeError):
        ops.siegel_dist_forward(torch.zeros(2, 2, 17, 17, device=dev), torch.zeros(2, 2, 17, 17, device=dev))


@pytest.mark.parametrize("model,metric,n,b,N", [("upper", "riem", 4, 8192, 1093), ("bounded", "finf", 4, 65536, 5041),
                                                ("bounded", "riem", 4, 65536, 5041), ("upper", "riem", 4, 65536, 5041),
                                                ("upper", "riem", 8, 262144, 45500), ("upper", "riem", 2, 512, 125)])
def test_full_size_properties(dev, model, metric, n, b, N):
    """BASELINE.json config sizes in their own metric (configs[2] is bounded / F-infinity; the headline shape upper / riem /
    n = 4 / 65 536 pairs / 5 041 rows is here too), checked through properties that need no CPU reference: symmetry
    d(x,y) = d(y,x); d(x,x) = 0; invariance under the isometries Z -> A Z A^T + S of the upper half space (and agreement
    upper == bounded through the Cayley map); plus a 256-pair sample against the oracle."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(11)
    tab_u = upper_points(N, n, 0.4, g)
    table = (tab_u if model == "upper" else __import__("tests.helpers", fromlist=["to_bounded"]).to_bounded(tab_u)).to(dev)
    src = torch.randint(0, N, (b,), generator=g)
    dst = (src + 1 + torch.randint(0, N - 1, (b,), generator=g)) % N
    trip = torch.stack((src, dst), 1).to(dev)
    d_xy = ops.model_forward(table, trip, model, metric)
    d_yx = ops.model_forward(table, trip.flip(1).contiguous(), model, metric)
    ops.check_status(dev)
    assert torch.isfinite(d_xy).all() and (d_xy > 0).all()
    assert rel_err(d_xy.cpu(), d_yx.cpu()) < 1e-10
    same = torch.stack((src, src), 1).to(dev)
    assert torch.all(ops.model_forward(table, same, model, metric) == 0)
    if model == "upper":
        a = (torch.eye(n) + 0.3 * torch.randn(n, n, generator=g)).to(dev)
        s = sym(torch.randn(n, n, generator=g)).to(dev)
        moved = torch.stack((a @ table[:, 0] @ a.T + s, a @ table[:, 1] @ a.T), 1)
        moved = torch.stack((sym(moved[:, 0]), sym(moved[:, 1])), 1)
        d_moved = ops.model_forward(moved, trip, model, metric)
        assert rel_err(d_moved.cpu(), d_xy.cpu()) < 1e-9
    else:
        d_up = ops.model_forward(tab_u.to(dev), trip, "upper", metric)
        assert rel_err(d_xy.cpu(), d_up.cpu()) < 1e-9
    # a 256-pair sample against the oracle as well
    k = 256
    want = so.model_forward(table.cpu(), trip[:k].cpu(), model, metric)
    assert rel_err(d_xy[:k].cpu(), want) < 1e-8


@pytest.mark.parametrize("dims", [4, 6, 10])
@pytest.mark.parametrize("model", MODELS)
def test_all_pairs_matrix_equals_runner_loop(dev, model, dims):
    """Model.distance_matrix == the reference's Runner.build_distance_matrix loop (runner.py:142-154):
    row i = forward of the pairs (i, j) for all j, self pair replaced and overwritten by 0."""
    from sympa_amd import ops
    from sympa_amd.model import Model

    class A:
        manifold, metric = model, "fone"
        scale_coef, scale_init, train_scale = 1.0, 1.7, False
    A.manifold = model
    A.dims = dims
    A.num_points = n_nodes = 131 if dims <= 6 else 71      # the CPU oracle loop dominates the test time
    g = torch.Generator().manual_seed(13)
    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = points(model, n_nodes, dims, 0.4 if dims <= 6 else 0.2, g)
    m = m.to(dev)
    with torch.no_grad():
        full = m.distance_matrix()
        block = m.distance_matrix(row_begin=17, row_count=40)
    ops.check_status(dev)
    want = torch.zeros(n_nodes, n_nodes, dtype=torch.float64)
    all_nodes = torch.arange(n_nodes).unsqueeze(1)
    for node in range(n_nodes):            # the reference loop, with the oracle as forward
        src = torch.full((n_nodes, 1), node)
        src[node] = (node + 1) % n_nodes
        d = so.model_forward(m.embeddings.embeds.detach().cpu(), torch.cat((src, all_nodes), -1), model, "fone",
                             scale=m.scale.detach().cpu(), scale_coef=1.0)
        d[node] = 0
        want[node] = d
    assert full.shape == (n_nodes, n_nodes) and torch.all(full.diagonal() == 0)
    assert rel_err(full.cpu(), want) < (TOL if dims <= 6 else 1e-7)
    # n >= 5: the lockstep QL makes the last bits depend on the pairs that share a wave; n <= 4: the full matrix stores
    # d(j, i) for the entries below the diagonal tiles (packed kernel), a row block evaluates (i, j) itself
    assert rel_err(block.cpu(), full[17:57].cpu()) < 1e-12
    assert rel_err(full.cpu(), full.cpu().T) < 1e-10


@pytest.mark.parametrize("n", [9, 12, 16])
@pytest.mark.parametrize("model", MODELS)
def test_generic_dims_fallback_kernel(dev, model, n):
    """dims 9..16 run the runtime-n fallback kernel (scratch-resident matrices): parity with the oracle."""
    g = torch.Generator().manual_seed(300 + n)
    z1, z2 = points(model, 200, n, 0.2, g), points(model, 200, n, 0.2, g)
    for metric in ("riem", "fmin"):
        got = gpu_dist(z1, z2, model, metric)
        assert rel_err(got, so.manifold_dist(model, z1, z2, metric)) < 1e-8, (model, n, metric)


@pytest.mark.parametrize("n", list(range(9, 17)))      # EVERY instantiation of the layout
@pytest.mark.parametrize("model", MODELS)
def test_dims_9_to_16_cooperative_kernel(dev, model, n):
    """9 <= n <= 16: sixteen lanes per pair (csrc/siegel_coop.hpp), n < 16 padded with the point i I (upper) / 0 (bounded).
    Against the oracle, against the runtime-n kernel (FLAG_GENERIC), every metric, the vector-valued distance, ragged
    batch sizes, the gathered form with scale, and the status word."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(1300 + n)
    w = torch.linspace(-0.4, 1.3, n, dtype=torch.float64)
    for b, s in ((1, 0.2), (65, 1e-3), (150, 0.2)):
        z1, z2 = points(model, b, n, s, g), points(model, b, n, s, g)
        for metric in METRICS:
            coop, vv = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric, w.to(dev), return_vvd=True)
            ops.check_status(dev)
            gen, vg = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric, w.to(dev), return_vvd=True,
                                              flags=ops.FLAG_GENERIC)
            ops.check_status(dev)
            want = so.manifold_dist(model, z1, z2, metric, w)
            # the oracle follows the reference's own chain (sqrt, inverses, 2n x 2n eigh): it carries ~1e-8 itself
            # (tests/golden *__vvd_exact50); the two kernels evaluate the same formula
            assert rel_err(coop.cpu(), want) < 1e-7, (n, b, s, metric)
            assert rel_err(coop.cpu(), gen.cpu()) < 1e-10, (n, b, s, metric)
            assert rel_err(vv.cpu(), vg.cpu()) < 1e-9 and vv.shape == (b, n)
    assert torch.all(ops.siegel_dist_forward(z1.to(dev), z1.to(dev), model, "riem") == 0)
    table = points(model, 90, n, 0.2, g)
    trip = torch.randint(0, 90, (333, 3), generator=g)
    scale = torch.tensor([1.7], dtype=torch.float64)
    out = ops.model_forward(table.to(dev), trip.to(dev), model, "fone", None, scale.to(dev), 2.0).cpu()
    ops.check_status(dev)
    assert rel_err(out, so.model_forward(table, trip, model, "fone", scale=scale, scale_coef=2.0)) < 1e-9
    bad = z1.clone()
    if model == "upper":
        bad[7, 1] = -bad[7, 1]               # Im z not positive definite
    else:
        bad[7] = 3.0 * bad[7] + torch.eye(n, dtype=torch.float64)        # outside the bounded domain
    ops.siegel_dist_forward(bad.to(dev), z2.to(dev), model, "riem")
    with pytest.raises(AssertionError):
        ops.check_status(dev)


@pytest.mark.parametrize("model,n", [("upper", 4), ("bounded", 4), ("upper", 8), ("bounded", 6), ("upper", 12)])
def test_triangle_inequality_of_the_true_metrics(dev, model, n):
    """riem (Riemannian), fone and finf (Finsler) are distances: d(x, z) <= d(x, y) + d(y, z) on random triples, over
    every kernel family (register n <= 4, QL n <= 8, sixteen lanes n >= 9)."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(4000 + n)
    b = 4096
    x, y, z = (points(model, b, n, 0.4 if n <= 8 else 0.2, g).to(dev) for _ in range(3))
    for metric in ("riem", "fone", "finf"):
        dxz = ops.siegel_dist_forward(x, z, model, metric)
        dxy = ops.siegel_dist_forward(x, y, model, metric)
        dyz = ops.siegel_dist_forward(y, z, model, metric)
        ops.check_status(dev)
        assert torch.all(dxz <= (dxy + dyz) * (1 + 1e-12)), (model, n, metric)
        assert torch.all(dxz > 0)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 10, 16])
@pytest.mark.parametrize("model", MODELS)
def test_nonfinite_input_gives_nan_and_status(dev, model, n):
    """NaN / Inf in a point: NaN out + ST_NONFINITE (reference: NaN eigenvalues fail the assert,
    siegel_manifold.py:64-66) -- not a distance of 0 through the eigenvalue clamp or a max / min metric."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(2000 + n)
    b = 130
    for bad in (float("nan"), float("inf")):
        for plane in (0, 1):
            z1, z2 = points(model, b, n, 0.3, g), points(model, b, n, 0.3, g)
            rows = [3, 64, 129]
            for k, r in enumerate(rows):
                tgt = z1 if k % 2 == 0 else z2
                i, j = (0, n - 1) if k < 2 else (n - 1, n - 1)
                tgt[r, plane, i, j] = bad
                tgt[r, plane, j, i] = bad
            for metric in METRICS:
                w = torch.linspace(0.2, 1.2, n).to(dev)
                out = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric, w).cpu()
                good = torch.ones(b, dtype=torch.bool)
                good[rows] = False
                assert torch.isnan(out[rows]).all(), (model, n, bad, plane, metric, out[rows])
                assert torch.isfinite(out[good]).all()
                with pytest.raises(AssertionError):
                    ops.check_status(dev)
    # through the fused gather too: a table row with a NaN real part (what rsgd leaves behind a NaN gradient)
    table = points(model, 50, n, 0.3, g)
    table[7, 0] = float("nan")
    trip = torch.tensor([[7, 1], [2, 7], [3, 4]])
    out = ops.model_forward(table.to(dev), trip.to(dev), model, "riem").cpu()
    assert torch.isnan(out[0]) and torch.isnan(out[1]) and torch.isfinite(out[2])
    with pytest.raises(AssertionError):
        ops.check_status(dev)


@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("model", MODELS)
def test_low_lds_gather_variants_bit_identical(dev, model, n):
    """The minimum-LDS gather forms (SYMPA_FLAG_LOW_LDS: PASS4 at n = 4, one endpoint at a time at n = 2; chosen
    automatically for grids deeper than two blocks per CU, and by bench.py's overlapped launches) run the same
    arithmetic on the same rows: outputs must be bit-identical to the default form."""
    from sympa_amd import data, ops
    N = 777
    table = data.trained_like_table(N, n, model=model, seed=11).to(dev)
    for b in (1, 63, 1000, 65536 + 37, 131072 + 4097):          # ragged; > 131 072 selects the low form by itself
        trip = data.sample_pairs(N, b, 3, 5).to(dev)
        base = ops.model_forward(table, trip, model, "riem", flags=0)
        low = ops.model_forward(table, trip, model, "riem", flags=ops.FLAG_LOW_LDS)
        ops.check_status(dev)
        assert torch.equal(low, base), (model, n, b)
        if b <= 1000:
            want = so.model_forward(table.cpu(), trip.cpu(), model, "riem")
            assert rel_err(low.cpu(), want) < TOL
        else:
            # explicit gather + pre-gathered entry (no index path at all) on a slice
            sl = slice(b - 5000, b)
            two = ops.siegel_dist_forward(table[trip[sl, 0]], table[trip[sl, 1]], model, "riem", flags=0)
            assert torch.equal(low[sl], two)
    # all-pairs matrix with N >= 363 rows (deep grid -> low form) against the pairwise kernel in its default form
    N2 = 400
    t2 = table[:N2].contiguous()
    mat = ops.all_pairs_dist(t2, model, "riem", packed=False, flags=ops.FLAG_NO_SYMMETRY)
    ii, jj = torch.meshgrid(torch.arange(N2, device=dev), torch.arange(N2, device=dev), indexing="ij")
    trip = torch.stack((ii.reshape(-1), jj.reshape(-1)), 1)
    chunks = [ops.model_forward(t2, trip[k:k + 60000].contiguous(), model, "riem", flags=0)
              for k in range(0, N2 * N2, 60000)]
    assert torch.equal(mat.reshape(-1), torch.cat(chunks))


def test_batched_forward_equals_single_calls(dev):
    """C-ABI sympa_model_forward_batches (the loop of Runner.evaluate, runner.py:126-137, in one call, launches
    alternating over streams) == one sympa_model_forward per batch, bit for bit; ragged batch sizes."""
    from sympa_amd import data, ops
    N, n = 500, 4
    table = data.trained_like_table(N, n, seed=3).to(dev)
    scale = torch.tensor([0.7], device=dev)
    sizes = [1, 300, 65536, 4099, 256, 70000]
    batches = [data.sample_pairs(N, b, k, 9).to(dev) for k, b in enumerate(sizes)]
    outs = [torch.zeros(b, dtype=torch.float64, device=dev) for b in sizes]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    for fl in (0, ops.FLAG_LOW_LDS, ops.FLAG_LOW_LDS | ops.FLAG_ANY_ORDER):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        bf = ops.BatchedForward(table, batches, outs, "upper", "fone", None, scale, 2.0, flags=fl, streams=streams)
        bf.run()
        torch.cuda.synchronize()
        ops.check_status(dev)
        for t, o in zip(batches, outs):
            assert torch.equal(o, ops.model_forward(table, t, "upper", "fone", None, scale, 2.0))
    bf.run(2, 2)          # a sub-range
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        bf.run(5, 2)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("model", MODELS)
def test_fused_batches_bit_identical(dev, model, n):
    """SYMPA_FLAG_FUSE: up to 32 batches per kernel launch (block -> batch by a prefix-sum search) == one launch per
    batch, bit for bit; ragged and empty batches, more than one group, bad index flagged in the right batch."""
    from sympa_amd import data, ops
    N = 300
    table = data.trained_like_table(N, n, model=model, seed=5).to(dev)
    scale = torch.tensor([1.3], device=dev)
    sizes = [257, 1, 0, 4096, 255, 256, 70001 if n <= 4 else 9001] + [100 + 37 * k for k in range(30)]
    batches = [data.sample_pairs(N, max(b, 1), k, 21)[:b].contiguous().to(dev) for k, b in enumerate(sizes)]
    outs = [torch.full((b,), -1.0, dtype=torch.float64, device=dev) for b in sizes]
    w = torch.linspace(0.1, 1.0, n).to(dev)
    bf = ops.BatchedForward(table, batches, outs, model, "wsum", w, scale, 1.5, flags=ops.FLAG_FUSE)
    assert len(sizes) > ops.MAX_FUSED_BATCHES
    bf.run()
    torch.cuda.synchronize()
    ops.check_status(dev)
    for t, o in zip(batches, outs):
        if t.shape[0]:
            assert torch.equal(o, ops.model_forward(table, t, model, "wsum", w, scale, 1.5))
    # an out-of-range index in batch 4 only
    batches[4][7, 1] = N
    bf2 = ops.BatchedForward(table, batches[:8], outs[:8], model, "riem", None, scale, 1.5, flags=ops.FLAG_FUSE)
    bf2.run()
    torch.cuda.synchronize()
    assert torch.isnan(outs[4][7]) and torch.isfinite(outs[4][:7]).all() and torch.isfinite(outs[3]).all()
    with pytest.raises(IndexError):
        ops.check_status(dev)


def test_integration_md_stub_runs_on_the_gpu(dev):
    """The ctypes stub of INTEGRATION.md section 3 (what a maintainer pastes into the reference's siegel_manifold.py),
    executed as written: forward and backward through the C-ABI equal the package's own binding."""
    import os
    import re
    from abc import ABC
    from sympa_amd import autograd as sa
    from sympa_amd.manifolds import metrics as mm
    from tests.helpers import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = re.findall(r"```python\n(.*?)```", text[text.index("## 3."):text.index("## 4.")], flags=re.S)[0]
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = {"Manifold": type("Manifold", (), {}), "ABC": ABC}   # geoopt's base class stands in
        exec(compile(src, "INTEGRATION.md#3", "exec"), ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(77)
    n = 3
    for model_id, model in enumerate(MODELS):
        for t in (mm.MetricType.RIEMANNIAN, mm.MetricType.WEIGHTED_SUM):
            man = ns["SiegelManifold"]()
            man.dims, man.model_id = n, model_id
            man.metric = mm.Metric.get(t, n)
            if t is mm.MetricType.WEIGHTED_SUM:
                man.metric = man.metric.to(dev)
            z1 = points(model, 50, n, 0.4, g).to(dev).requires_grad_(True)
            z2 = points(model, 50, n, 0.4, g).to(dev).requires_grad_(True)
            d = man.dist(z1, z2)
            assert man.dist(z1, z2, keepdim=True).shape == (50, 1)
            d.sum().backward()
            w = getattr(man.metric, "weights", None)
            y1, y2 = z1.detach().clone().requires_grad_(True), z2.detach().clone().requires_grad_(True)
            ref = sa.siegel_dist(y1, y2, model, t.value, w)
            ref.sum().backward()
            assert torch.equal(d.detach(), ref.detach())
            assert torch.equal(z1.grad, y1.grad) and torch.equal(z2.grad, y2.grad)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_all_pairs_packed_kernel(dev, model, n):
    """sympa_all_pairs_dist_packed (every point factored once, inverted factor, wave-uniform row point, symmetric full
    matrix) against the pairwise kernel in index-free mode and against the oracle; ragged N, row blocks at odd offsets."""
    from sympa_amd import data, ops
    N = 333
    table = data.trained_like_table(N, n, model=model, seed=17).to(dev)
    scale = torch.tensor([0.8], device=dev)
    w = torch.linspace(-0.2, 1.1, n).to(dev)
    for metric in METRICS:
        full = ops.all_pairs_dist(table, model, metric, w, scale, 2.0, packed=True)
        ref = ops.all_pairs_dist(table, model, metric, w, scale, 2.0, packed=False, flags=ops.FLAG_NO_SYMMETRY)
        ops.check_status(dev)
        assert torch.all(full.diagonal() == 0)
        if n >= 3:
            assert torch.equal(full, full.T)                 # symmetric mode: d(i, j) evaluated once, stored twice
            both = ops.all_pairs_dist(table, model, metric, w, scale, 2.0, packed=True, flags=ops.FLAG_NO_SYMMETRY)
            assert rel_err(both.cpu(), ref.cpu()) < (1e-11 if n <= 4 else 1e-10) and torch.all(both.diagonal() == 0)
        # symmetric mode stores d(i, j) also at (j, i), where the pairwise kernel evaluates d(j, i): two runs of the
        # Jacobi iteration on different matrices with the same spectrum (measured agreement ~3e-11 at n = 4)
        assert rel_err(full.cpu(), ref.cpu()) < (2e-10 if n >= 3 else 1e-11), (model, n, metric)
        blk_tol = 1e-11 if n <= 4 else 1e-10          # n >= 5: the lockstep QL makes the last bits depend on which pairs share a wave
        for rb, rc in ((0, 1), (5, 64), (63, 66), (100, 233), (332, 1)):
            blk = ops.all_pairs_dist(table, model, metric, w, scale, 2.0, row_begin=rb, row_count=rc, packed=True)
            assert blk.shape == (rc, N)
            assert rel_err(blk.cpu(), ref[rb:rb + rc].cpu()) < blk_tol, (model, n, metric, rb, rc)
    ii, jj = torch.meshgrid(torch.arange(40), torch.arange(N), indexing="ij")
    want = so.model_forward(table.cpu(), torch.stack((ii.reshape(-1), jj.reshape(-1)), 1), model, "riem",
                            scale=scale.cpu(), scale_coef=2.0).reshape(40, N)
    got = ops.all_pairs_dist(table, model, "riem", None, scale, 2.0, packed=True)[:40].cpu()
    assert rel_err(got, want, atol=1e-9) < TOL
    # a point outside the manifold is flagged by the pack kernel
    bad = table.clone()
    bad[7, 1] = -bad[7, 1] if model == "upper" else 3.0 * bad[7, 1] + 2.0
    ops.all_pairs_dist(bad, model, "riem", packed=True)
    with pytest.raises(AssertionError):
        ops.check_status(dev)


@pytest.mark.parametrize("n", [3, 5, 8, 10])
@pytest.mark.parametrize("model", MODELS)
def test_all_pairs_symmetric_mode_of_the_pairwise_kernels(dev, model, n):
    """Full matrix through the pairwise kernels (dims >= 5; `packed=False` below that): only the pairs i <= j are
    evaluated (triangular pair index inverted with one square root per lane), every value stored at (i, j) and (j, i).
    Against the both-orders evaluation; ragged N around the integer-square-root edge cases."""
    from sympa_amd import data, ops
    for N in (1, 2, 63, 64, 65, 257, 700):
        table = data.trained_like_table(max(N, 2), n, model=model, seed=23, scale=0.2)[:N].contiguous().to(dev)
        sym = ops.all_pairs_dist(table, model, "fone", packed=False)
        both = ops.all_pairs_dist(table, model, "fone", packed=False, flags=ops.FLAG_NO_SYMMETRY)
        ops.check_status(dev)
        assert sym.shape == (N, N) and torch.all(sym.diagonal() == 0) and torch.equal(sym, sym.T)
        assert rel_err(sym.cpu(), both.cpu()) < 1e-9, (model, n, N)
        iu = torch.triu_indices(N, N)
        if n <= 4:       # the i <= j entries are the same evaluation (n >= 5: the lockstep QL makes the last bits depend
            assert torch.equal(sym[iu[0], iu[1]], both[iu[0], iu[1]])      # on which pairs share a wave)
        else:
            assert rel_err(sym[iu[0], iu[1]].cpu(), both[iu[0], iu[1]].cpu()) < 1e-11


def test_fused_multi_batch_kernel_against_the_reference_golden_directly(dev):
    """The headline kernel (siegel_dist_multi_kernel behind SYMPA_FLAG_FUSE) against the imported reference's
    Model.forward outputs themselves, not only through its bit-identity with the single-step kernel: the golden batch
    cut into ragged pieces that one fused launch evaluates together."""
    from sympa_amd import ops
    g = np.load(f"{GOLDEN}/model_forward.npz")
    trip = torch.from_numpy(g["triplets"]).to(dev)
    b = trip.shape[0]
    cuts = [0, 1, 7, b // 3, b // 3 + 65, b]
    for model in MODELS:
        table = T(g[f"table_{model}"]).to(dev)
        for scale, coef in ((1.0, 1.0), (3.0, 2.0)):
            sc = torch.tensor([scale], device=dev)
            batches = [trip[a:e].contiguous() for a, e in zip(cuts[:-1], cuts[1:])]
            outs = [torch.empty(t.shape[0], dtype=torch.float64, device=dev) for t in batches]
            ops.BatchedForward(table, batches, outs, model, "riem", None, sc, coef, flags=ops.FLAG_FUSE).run()
            ops.check_status(dev)
            assert rel_err(torch.cat(outs).cpu(), g[f"{model}__scale{scale}_coef{coef}"]) < TOL


def _model(manifold, metric, dims, num_points, dev, table=None, scale_init=1.0, scale_coef=1.0):
    from sympa_amd.model import Model

    class A:
        pass
    A.manifold, A.metric, A.dims, A.num_points = manifold, metric, dims, num_points
    A.scale_coef, A.scale_init, A.train_scale = scale_coef, scale_init, False
    m = Model(A)
    if table is not None:
        with torch.no_grad():
            m.embeddings.embeds.data = table.clone()
    return m.to(dev)


@pytest.mark.parametrize("model,metric,n", [("upper", "riem", 4), ("bounded", "finf", 4), ("upper", "wsum", 3),
                                            ("upper", "riem", 8), ("bounded", "fone", 10), ("spd", "riem", 5)])
def test_model_forward_batches_equals_one_forward_per_batch(dev, model, metric, n):
    """Model.forward_batches (one C call, fused multi-batch launches for dims <= 8) == the loop of Runner.evaluate
    (runner.py:126-131: one forward() per batch), bit for bit for n <= 4; list form, plan form, cached plan, outputs
    handed in, empty and ragged batches."""
    from sympa_amd import data, ops
    N = 300
    g = torch.Generator().manual_seed(11)
    table = data.spd_table(N, n, seed=3) if model == "spd" else points(model, N, n, 0.4, g)
    m = _model(model, metric, n, N, dev, table, scale_init=1.7, scale_coef=2.0)
    sizes = [257, 0, 1, 64, 1000, 33] + [97] * 40          # 46 batches: two fused launches (32 + 14)
    batches = [torch.randint(0, N, (s, 3), generator=g).to(dev) for s in sizes]
    with torch.no_grad():
        want = [m(t) for t in batches]
    got = m.forward_batches(batches)
    ops.check_status(dev)
    exact = model != "spd" and n <= 4
    for a, b_ in zip(got, want):
        assert a.shape == b_.shape
        assert torch.equal(a, b_) if exact else torch.allclose(a, b_, rtol=1e-11, atol=1e-13)
    assert m.forward_batches(batches)[0].data_ptr() == got[0].data_ptr()          # cached plan, same outputs
    plan = m.prepare_batches(batches[:5])
    got2 = m.forward_batches(plan)
    assert all(torch.equal(a, b_) if exact else torch.allclose(a, b_, rtol=1e-11, atol=1e-13)
               for a, b_ in zip(got2, want[:5]))
    outs = [torch.full((s,), -1.0, dtype=torch.float64, device=dev) for s in sizes]
    got3 = m.forward_batches(batches, outs)
    assert all(a.data_ptr() == o.data_ptr() for a, o in zip(got3, outs) if o.numel())
    assert all((o >= 0).all() for o in outs)
    # the parameters are read through their device pointers: an in-place update is seen by a cached plan
    with torch.no_grad():
        m.scale.mul_(2.0)
        want2 = m(batches[4])
    assert torch.allclose(m.forward_batches(batches)[4], want2, rtol=1e-12, atol=0)


def test_model_evaluate_equals_the_reference_loop_on_configs1_triplets(dev):
    """Model.evaluate == Runner.evaluate (runner.py:124-135) with the oracle as the model, on every one of the 596 778
    (i < j, d) triplets of configs[1]'s graph (balanced tree b = 3, h = 6; batch 8 192): per-triplet distortions
    |d_m - d_g| / d_g (metrics.py:21), their mean, and the per-triplet distances themselves."""
    from statistics import mean
    from sympa_amd import data, ops
    trip, id2node = data.graph_triplets(data.named_graph("tree-b3-h6"))
    assert trip.shape == (596778, 3)
    N, n, batch = len(id2node), 4, 8192
    table = data.trained_like_table(N, n, seed=5)
    m = _model("upper", "riem", n, N, dev, table)
    ids, gd = trip[:, :2].contiguous().to(dev), trip[:, 2].to(torch.float64).to(dev)
    got = m.evaluate(ids, gd, batch)
    ops.check_status(dev)
    dists = m._eval_plan[2].cpu()
    # the reference's loop: forward per batch, extend a python list, statistics.mean
    total = []
    torch.set_num_threads(8)
    want_d = torch.empty(trip.shape[0], dtype=torch.float64)
    for s in range(0, trip.shape[0], 65536):
        want_d[s:s + 65536] = so.model_forward(table, trip[s:s + 65536], "upper", "riem")
    for s in range(0, trip.shape[0], batch):
        d = want_d[s:s + batch]
        gdb = trip[s:s + batch, 2].to(torch.float64)
        total.extend((torch.abs(d - gdb) / gdb).tolist())
    assert rel_err(dists, want_d) < TOL
    assert abs(got - mean(total)) < 1e-12 * max(1.0, mean(total))
    # second call: cached plan, same answer; [T,3] triplets work as well
    assert m.evaluate(ids, gd, batch) == got
    assert abs(m.evaluate(trip.to(dev), gd, batch) - got) < 1e-15


def test_selfcheck_gate_passes_for_every_instantiation_and_falls_back_when_told(dev):
    """sympa_amd/selfcheck.py: every (family, model, n) of the lanes-per-pair layout agrees with the one-lane kernel on this
    build (the gate that runs on first use in production); and an instantiation marked through the C-ABI registry really
    is served by the one-lane kernel (same numbers as FLAG_GENERIC, bit for bit)."""
    from sympa_amd import _lib, ops, selfcheck
    assert selfcheck.ENABLED
    failures = selfcheck.check_all(dev)
    assert failures == [], failures
    assert len(selfcheck.CHECKED) >= 2 * 8 + 2 * 10 + 2 * 10 + 11 + 14 + 14
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    z1, z2 = points("upper", 130, 11, 0.3, g).to(dev), points("upper", 130, 11, 0.3, g).to(dev)
    fast = ops.siegel_dist_forward(z1, z2)
    generic = ops.siegel_dist_forward(z1, z2, flags=ops.FLAG_GENERIC)
    assert lib.sympa_set_instance_fallback(selfcheck.SIEGEL_FWD, 0, 11, 1) == 0
    try:
        assert lib.sympa_get_instance_fallback(selfcheck.SIEGEL_FWD, 0, 11) == 1
        routed = ops.siegel_dist_forward(z1, z2)
    finally:
        lib.sympa_set_instance_fallback(selfcheck.SIEGEL_FWD, 0, 11, 0)
    assert torch.equal(routed, generic) and rel_err(fast.cpu(), generic.cpu()) < 1e-10
    assert lib.sympa_set_instance_fallback(9, 0, 11, 1) != 0          # no such family
    # spd: a routed backward-with-scatter goes through rows + scatter-add
    from sympa_amd import data
    table = data.spd_table(60, 7, scale=0.3, seed=3).to(dev)
    trip = torch.randint(0, 60, (500, 2), generator=g).to(dev)
    gd = torch.ones(500, dtype=torch.float64, device=dev)
    a, b_ = torch.zeros_like(table), torch.zeros_like(table)
    la, lb = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    ops.spd_loss_backward(table, trip, a, graph_dist=gd, loss=la)
    lib.sympa_set_instance_fallback(selfcheck.SPD_BWD, 0, 7, 1)
    try:
        ops.spd_loss_backward(table, trip, b_, graph_dist=gd, loss=lb)
    finally:
        lib.sympa_set_instance_fallback(selfcheck.SPD_BWD, 0, 7, 0)
    assert float((a - b_).abs().max()) < 1e-9 * float(a.abs().max()) and abs(float(la - lb)) < 1e-9 * float(la)


@pytest.mark.parametrize("n", [7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_eight_lanes_per_pair_forward_ab_kernel(dev, model, n):
    """SYMPA_FLAG_COOP at dims 7, 8: the forward with eight lanes per pair (two pairs per DPP row; csrc/siegel_coop_half.hip),
    kept for the A/B against the one-pair-per-lane register kernels: same distances, ragged batches, the gathered form."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(1700 + n)
    for b, s in ((1, 0.3), (67, 1e-3), (1000, 0.3), (4099, 0.6)):
        z1, z2 = points(model, b, n, s, g), points(model, b, n, s, g)
        for metric in ("riem", "fmin"):
            a = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric, flags=ops.FLAG_COOP).cpu()
            c = ops.siegel_dist_forward(z1.to(dev), z2.to(dev), model, metric).cpu()
            ops.check_status(dev)
            assert rel_err(a, c) < 1e-10, (model, n, b, metric)
            assert rel_err(a, so.manifold_dist(model, z1, z2, metric)) < 1e-8
    table = points(model, 300, n, 0.3, g).to(dev)
    trip = torch.randint(0, 300, (2500, 3), generator=g).to(dev)
    sc = torch.tensor([1.3], device=dev)
    a = ops.model_forward(table, trip, model, "riem", None, sc, 2.0, flags=ops.FLAG_COOP)
    c = ops.model_forward(table, trip, model, "riem", None, sc, 2.0)
    assert rel_err(a.cpu(), c.cpu()) < 1e-10


@pytest.mark.gpu
def test_thin_torch_binding_equals_the_ctypes_binding(dev):
    """sympa_amd/_fast (csrc/torch_binding.cpp) is a second binding of C-ABI sympa_model_forward for per-batch callers: bit-
    identical outputs to the ctypes route for every model / metric / stride form, on the CURRENT stream (a side stream's
    launch is ordered after that stream's work), same error types, and it is the route Model.forward takes."""
    from sympa_amd import _lib, ops
    assert _lib.fast() is not None, "sympa_amd/_fast is not built (__graft_entry__.build())"
    g = torch.Generator().manual_seed(3)
    for model, n in (("upper", 4), ("bounded", 3), ("upper", 8), ("upper", 11)):
        tab_u = upper_points(120, n, 0.4, g)
        table = (tab_u if model == "upper" else __import__("tests.helpers", fromlist=["to_bounded"]).to_bounded(tab_u)).to(dev)
        trip3 = torch.stack((torch.randint(0, 120, (700,), generator=g), torch.randint(0, 120, (700,), generator=g),
                             torch.randint(1, 9, (700,), generator=g)), 1).to(dev)
        w = torch.linspace(0.2, 1.4, n, dtype=torch.float64).to(dev)
        sc = torch.tensor([1.3], dtype=torch.float64, device=dev)
        for metric in ("riem", "finf", "wsum"):
            for trip in (trip3, trip3[:, :2].contiguous(), trip3[:1]):
                _lib._fast = None
                a = ops.model_forward(table, trip, model, metric, w, sc, 2.0)
                _lib._fast = False                    # ctypes
                b = ops.model_forward(table, trip, model, metric, w, sc, 2.0)
                _lib._fast = None
                assert torch.equal(a, b), (model, n, metric)
    # stream semantics: on a side stream the launch is ordered behind that stream's earlier work
    side = torch.cuda.Stream()
    table = upper_points(120, 4, 0.4, g).to(dev)
    trip = torch.stack((torch.randint(0, 120, (4096,), generator=g), torch.randint(0, 120, (4096,), generator=g)), 1).to(dev)
    want = ops.model_forward(table, trip)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        moved = torch.zeros_like(trip)
        torch.cuda._sleep(20_000_000)
        moved.copy_(trip)                             # the forward must see the copied indices
        got = ops.model_forward(table, moved)
    side.synchronize()
    assert torch.equal(got, want)
    # error contract
    with pytest.raises(_lib.SympaHipError):
        ops.model_forward(table.cpu(), trip.cpu())
    with pytest.raises(TypeError):
        ops.model_forward(table.float(), trip)
    with pytest.raises(RuntimeError):
        ops.model_forward(torch.zeros(4, 2, 17, 17, dtype=torch.float64, device=dev), trip)
    ops.model_forward(table, torch.tensor([[0, 500]], device=dev))
    with pytest.raises(IndexError):
        ops.check_status(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 8])
def test_forward_on_graded_spectra_against_the_oracle(dev, n):
    """tests/test_hostsim_parity.py::test_forward_on_graded_spectra_against_the_oracle through the HIP kernels (n = 8: the persistent
    dense kernel and, over a packed table, the packed kernel)."""
    from oracle import siegel_oracle as so
    from sympa_amd import ops
    from tests.helpers import graded_pairs
    for grade, tol in ((2, 1e-13), (4, 1e-11), (6, 1e-9), (8, 1e-7)):
        z1, z2 = graded_pairs(70, n, grade)
        t1, t2 = torch.from_numpy(z1), torch.from_numpy(z2)
        for metric in ("riem", "fone", "fmin"):
            ref = so.manifold_dist("upper", t1, t2, metric, None, False)
            got = ops.siegel_dist_forward(t1.to(dev), t2.to(dev), "upper", metric).cpu()
            assert float((got - ref).abs().max() / ref.abs().max()) < tol, (n, grade, metric)
            if n == 8:
                table = torch.cat((t1, t2)).to(dev)
                trip = torch.stack((torch.arange(70), torch.arange(70) + 70), 1).to(dev)
                pk = ops.PackedTable("upper").ensure(table)
                got = ops.model_forward_packed(pk, trip, metric).cpu()
                assert float((got - ref).abs().max() / ref.abs().max()) < tol, (n, grade, metric, "packed")
    ops.check_status(dev)
