"""CPU (`-m "not gpu"`): the per-pair arithmetic the gfx950 kernel runs (siegel_math.hpp compiled by
g++, tests/hostsim) against the golden vectors of the imported reference and against the oracle.
Tolerance: 1e-9 relative (abs floor 1e-9 for d ~ 0); north_star asks 1e-4."""
import numpy as np
import pytest
import torch

from oracle import siegel_oracle as so
from tests.helpers import GOLDEN, METRICS, MODELS, T, graded_pairs, hostsim_dist, points, rel_err

TOL = 1e-9
# 'far' (and the tail of 's1.0' at n=8) = the regime 1 - d in [1e-8, 1e-5] where the reference's own fp64 evaluation carries ~1e-7
# error (see vvd_exact50); there we hold 1e-6 against the reference and 1e-12 against the exact value.
TOL_FAR_VS_REFERENCE = 1e-6
TOL_FAR_REFERENCE_LARGE_N = 1e-4     # dims 5..8 (tests/golden/dist_far_*): see test_golden_far


def out_riem(z1, z2, model):
    return hostsim_dist(z1, z2, model, "riem")[0]


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_golden(model, n):
    g = np.load(f"{GOLDEN}/dist_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        for metric in METRICS:
            out, vvd, st = hostsim_dist(z1, z2, model, metric, g["wsum_weights"])
            assert st == 0
            tol = TOL_FAR_VS_REFERENCE if case in ("far", "s1.0") else TOL
            assert rel_err(out, g[f"{case}__{metric}"]) < tol, (model, n, case, metric)
        if f"{case}__vvd_exact50" in g:
            # bounded points near the boundary: forming I - W W^H cancels ~5 digits (conditioning of
            # the disc representation itself), so 1e-9 there; upper model: 1e-12
            tol_exact = 1e-12 if model == "upper" else 1e-9
            assert rel_err(vvd, g[f"{case}__vvd_exact50"]) < tol_exact, (model, n, case)


@pytest.mark.parametrize("n", [5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_golden_far(model, n):
    """Round 6: the reference's `far` (clamp-regime) outputs at dims 5..8 and their 50-digit evaluation."""
    g = np.load(f"{GOLDEN}/dist_far_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        exact = g[f"{case}__vvd_exact50"]
        for metric in METRICS:
            out, vvd, st = hostsim_dist(z1, z2, model, metric, g["wsum_weights"])
            assert st == 0
            # against the reference: north_star's 1e-4.  What is left IS the reference's own fp64 error in this regime: its riem
            # against the 50-digit evaluation of its own formula reaches 8.8e-5 at n = 7, 8 (2.2e-6 at n = 6) ...
            assert rel_err(out, g[f"{case}__{metric}"]) < TOL_FAR_REFERENCE_LARGE_N, (model, n, case, metric)
        # ... while the kernels' arithmetic holds 1e-8 against that evaluation (QL is accurate to eps * lambda_max absolutely: the
        # smallest of eigenvalues spread over 1e10 keeps ~9 digits; Jacobi at n <= 4 keeps full relative accuracy)
        assert rel_err(vvd, exact) < 1e-8, (model, n, case)
        assert rel_err(out_riem(z1, z2, model), np.sqrt((exact ** 2).sum(1))) < 1e-9, (model, n, case)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_oracle_seeded(model, n):
    g = torch.Generator().manual_seed(100 + n)
    for s in (1e-3, 0.3, 0.8):
        z1, z2 = points(model, 40, n, s, g), points(model, 40, n, s, g)
        for metric in ("riem", "fmin"):
            out, vvd, st = hostsim_dist(z1.numpy(), z2.numpy(), model, metric)
            want = so.manifold_dist(model, z1, z2, metric)
            assert st == 0
            assert rel_err(out, want) < TOL, (model, n, s, metric)
        v_ref, _ = so.vector_valued_distance(*( (so.inverse_cayley_transform(z1), so.inverse_cayley_transform(z2))
                                               if model == "bounded" else (z1, z2)))
        assert rel_err(vvd, v_ref, atol=1e-9) < 1e-7   # individual small v_i next to large ones


def test_status_not_positive_definite():
    g = torch.Generator().manual_seed(3)
    z1, z2 = points("upper", 8, 3, 0.3, g), points("upper", 8, 3, 0.3, g)
    z1[2, 1] = -z1[2, 1]     # Im z not PD
    _, _, st = hostsim_dist(z1.numpy(), z2.numpy(), "upper", "riem")
    assert st & 1


def test_same_point_is_exactly_zero():
    g = torch.Generator().manual_seed(4)
    for model in MODELS:
        z = points(model, 16, 4, 0.5, g)
        out, _, st = hostsim_dist(z.numpy(), z.numpy(), model, "riem")
        assert st == 0 and np.all(out == 0.0)


@pytest.mark.parametrize("n", [3, 8, 11, 16])
@pytest.mark.parametrize("model", MODELS)
def test_generic_runtime_n_fallback(model, n):
    """The runtime-n fallback (dims 9..16 on the GPU) against the oracle, and against the specialised code
    where both exist."""
    g = torch.Generator().manual_seed(200 + n)
    z1, z2 = points(model, 12, n, 0.25, g), points(model, 12, n, 0.25, g)
    for metric in ("riem", "finf", "wsum"):
        w = torch.linspace(-0.3, 1.2, n)
        out, vvd, st = hostsim_dist(z1.numpy(), z2.numpy(), model, metric, w.numpy(), generic=True)
        assert st == 0
        assert rel_err(out, so.manifold_dist(model, z1, z2, metric, w)) < 1e-8, (model, n, metric)
        if n <= 8:
            out2, _, _ = hostsim_dist(z1.numpy(), z2.numpy(), model, metric, w.numpy())
            assert rel_err(out, out2) < 1e-10


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("model", MODELS)
def test_nonfinite_input_gives_nan_and_status(model, n):
    """A NaN / Inf anywhere in a point must come out as NaN with ST_NONFINITE, for every metric (the reference
    produces NaN and fails `assert torch.all(eigvalues >= 0 - eps)`, siegel_manifold.py:64-66) -- never as a
    distance of 0 through the clamp of the eigenvalues at 0 or through a max / min reduction."""
    g = torch.Generator().manual_seed(300 + n)
    for bad in (float("nan"), float("inf"), -float("inf")):
        for plane in (0, 1):
            for which in (0, 1):
                for (i, j) in {(0, 0), (0, n - 1), (n - 1, n - 1)}:
                    z = [points(model, 1, n, 0.3, g), points(model, 1, n, 0.3, g)]
                    z[which][0, plane, i, j] = bad
                    z[which][0, plane, j, i] = bad
                    for metric in METRICS:
                        for generic in (False, True):
                            out, vvd, st = hostsim_dist(z[0].numpy(), z[1].numpy(), model, metric, generic=generic)
                            assert np.isnan(out[0]), (model, n, bad, plane, which, (i, j), metric, generic, out)
                            assert st & 2, (model, n, bad, plane, which, (i, j), metric, generic, st)
                            assert np.all(np.isnan(vvd))


@pytest.mark.parametrize("n", [11, 16])
def test_nonfinite_input_generic_large_n(n):
    g = torch.Generator().manual_seed(400 + n)
    for model in MODELS:
        z1, z2 = points(model, 1, n, 0.2, g), points(model, 1, n, 0.2, g)
        z2[0, 0, 2, 5] = float("nan")
        for metric in ("riem", "finf", "fmin"):
            out, _, st = hostsim_dist(z1.numpy(), z2.numpy(), model, metric, generic=True)
            assert np.isnan(out[0]) and (st & 2)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_packed_point_path_of_the_all_pairs_kernel(model, n):
    """All-pairs matrix (runner.py:142-154): points are packed once with their INVERTED Cholesky factor and a pair costs
    E = A1 (Z2 - Z1) A2^T.  Same distances as the solve-based pairwise arithmetic and as the oracle; goldens too."""
    from tests.helpers import hostsim_dist_packed
    g = torch.Generator().manual_seed(500 + n)
    for s in (1e-3, 0.3, 0.8):
        z1, z2 = points(model, 60, n, s, g), points(model, 60, n, s, g)
        z2[:5] = z1[:5]                                    # d(x, x) = 0 exactly
        for metric in ("riem", "finf", "wsum"):
            w = torch.linspace(-0.3, 1.2, n)
            got, st = hostsim_dist_packed(z1.numpy(), z2.numpy(), model, metric, w.numpy())
            ref, _, _ = hostsim_dist(z1.numpy(), z2.numpy(), model, metric, w.numpy())
            assert st == 0 and np.all(got[:5] == 0.0)
            assert rel_err(got, ref) < 1e-10, (model, n, s, metric)
            assert rel_err(got, so.manifold_dist(model, z1, z2, metric, w)) < TOL
            # the indexed packed forward's in-place difference form (dims 5..8 on the GPU; every n here)
            got_d, st_d = hostsim_dist_packed(z1.numpy(), z2.numpy(), model, metric, w.numpy(), diff=True)
            assert st_d == 0 and np.all(got_d[:5] == 0.0) and rel_err(got_d, got) < 1e-12, (model, n, s, metric)
    if n in (2, 3, 4, 5, 6, 7, 8):
        gold = np.load(f"{GOLDEN}/dist_{model}_n{n}.npz")
        for case in gold["case_names"]:
            tol = TOL_FAR_VS_REFERENCE if case in ("far", "s1.0") else TOL
            for diff in (False, True):
                got, st = hostsim_dist_packed(gold[f"{case}__z1"], gold[f"{case}__z2"], model, "riem", diff=diff)
                assert st == 0 and rel_err(got, gold[f"{case}__riem"]) < tol, (model, n, case, diff)


@pytest.mark.parametrize("n", [12, 16])
@pytest.mark.parametrize("model", MODELS)
def test_generic_arithmetic_against_reference_goldens(model, n):
    """The runtime-n arithmetic (the one-lane kernels behind SYMPA_FLAG_GENERIC, which the sixteen-lanes kernels are checked
    against on the GPU) against outputs of the imported reference at dims 12 and 16."""
    g = np.load(f"{GOLDEN}/dist_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        for metric in METRICS:
            out, vvd, st = hostsim_dist(z1, z2, model, metric, g["wsum_weights"], generic=True)
            assert st == 0
            tol = TOL_FAR_VS_REFERENCE if case in ("far", "s1.0") else TOL
            # d(x, x): exactly 0 here, ~1e-15 per component in the reference (up to 2(n-1) n of them under fmin)
            atol = 1e-10 if case == "same" else 1e-12
            assert rel_err(out, g[f"{case}__{metric}"], atol=atol) < tol, (model, n, case, metric)


@pytest.mark.parametrize("n", [4, 6, 8])
def test_forward_on_graded_spectra_against_the_oracle(n):
    """Pairs whose sinh^2(v_i / 2) spread over 1e-4 .. 1e-16 (tests/helpers.py::graded_pairs): the kernels take the eigenvalues of
    H = E^H E, the reference (and the oracle) the singular values of the Cayley image through a 2n x 2n symeig -- different routes to
    the small v_i.  Measured 1e-15 at a spread of 1e-4, 1e-8 at 1e-16; the tolerance of the path is 1e-4."""
    for grade, tol in ((2, 1e-13), (4, 1e-11), (6, 1e-9), (8, 1e-7)):
        z1, z2 = graded_pairs(12, n, grade)
        for metric in ("riem", "fone", "fmin", "finf"):
            got, st = hostsim_dist(z1, z2, "upper", metric)[:2]
            ref = so.manifold_dist("upper", torch.from_numpy(z1), torch.from_numpy(z2), metric, None, False).numpy()
            assert np.abs(got - ref).max() / np.abs(ref).max() < tol, (n, grade, metric)
