"""CPU: the eigenvector stage of the three-kernel SPD backward (sympa_amd/csrc/tridiag_invit.hpp, compiled by g++): eigenvalues
from the lockstep QL, eigenvectors by inverse iteration one matrix per lane -- against numpy.linalg.eigh on the spectra the
kernel meets (the bench table's tridiagonal forms, the init distribution, clusters of every tightness) and on the ones it
must hand back (blocks of more than INVIT_KEEP + 1 close eigenvalues)."""
import numpy as np
import pytest

from tests.helpers import hostsim_tridiag_invit


def tridiagonalize(a):
    """Householder tridiagonalisation (numpy), batch [b, n, n] -> d [b, n], e [b, n] (last column zero)."""
    a = a.copy()
    b, n, _ = a.shape
    for k in range(n - 2):
        x = a[:, k + 1:, k].copy()
        nx = np.linalg.norm(x, axis=1)
        v = x.copy()
        v[:, 0] += np.copysign(nx, x[:, 0])
        den = (v * v).sum(1)
        beta = np.where(den > 0, 2.0 / np.where(den > 0, den, 1.0), 0.0)
        sub = a[:, k + 1:, k + 1:]
        p = beta[:, None] * np.einsum("bij,bj->bi", sub, v)
        p -= (0.5 * beta * (v * p).sum(1))[:, None] * v
        sub -= v[:, :, None] * p[:, None, :] + p[:, :, None] * v[:, None, :]
        a[:, k + 1, k] = -np.copysign(nx, x[:, 0])
        a[:, k, k + 1] = a[:, k + 1, k]
        a[:, k + 2:, k] = 0
        a[:, k, k + 2:] = 0
    d = np.einsum("bii->bi", a).copy()
    e = np.zeros((b, n))
    for i in range(n - 1):
        e[:, i] = a[:, i + 1, i]
    return d, e


def dense(d, e):
    b, s = d.shape
    t = np.zeros((b, s, s))
    for i in range(s):
        t[:, i, i] = d[:, i]
    for i in range(s - 1):
        t[:, i + 1, i] = e[:, i]
        t[:, i, i + 1] = e[:, i]
    return t


def check(d, e, want_flagged=False):
    lam, z, flag = hostsim_tridiag_invit(d, e)
    t = dense(d, e)
    s = d.shape[1]
    nrm = np.maximum(np.abs(t).sum(1).max(1), 1e-300)
    assert (flag & 2 == 0).all()                                            # QL converged
    assert np.abs(lam - np.linalg.eigvalsh(t)).max() <= 1e-14 * nrm.max()
    if want_flagged:
        assert (flag & 1 == 1).all()
        return
    ok = flag == 0
    assert ok.mean() > 0.99
    z, t, lam, nrm = z[ok], t[ok], lam[ok], nrm[ok]
    assert np.abs(z @ np.swapaxes(z, 1, 2) - np.eye(s)).max() < 5e-13                     # orthonormal rows
    assert (np.abs(z @ t - lam[:, :, None] * z).max((1, 2)) / nrm).max() < 1e-14          # residual
    # what the backward needs: a smooth matrix function through the decomposition
    ww, vv = np.linalg.eigh(t)
    g = lambda x: np.log1p(np.maximum(x, -0.99))           # noqa: E731
    want = (vv * g(ww)[:, None, :]) @ np.swapaxes(vv, 1, 2)
    got = np.swapaxes(z, 1, 2) @ (g(lam)[:, :, None] * z)
    assert (np.abs(want - got).max((1, 2)) / np.maximum(np.abs(want).max((1, 2)), 1e-300)).max() < 1e-12


def test_bench_table_forms():
    """The tridiagonal forms of L^-1 (Y - X) L^-T on configs[4]'s table (keyed RNG: the bytes the bench uses)."""
    from sympa_amd import data
    tab = data.spd_table(3000, 16, seed=42).numpy()
    pr = data.sample_pairs(3000, 4096, 0, 42).numpy()
    x, y = tab[pr[:, 0]], tab[pr[:, 1]]
    li = np.linalg.inv(np.linalg.cholesky(x))
    m = li @ (y - x) @ np.swapaxes(li, -1, -2)
    check(*tridiagonalize(0.5 * (m + np.swapaxes(m, -1, -2))))


@pytest.mark.parametrize("s", [3, 8, 12, 16])
def test_random_spectra(s):
    rng = np.random.default_rng(s)
    a = rng.standard_normal((1500, s, s))
    check(*tridiagonalize((a + np.swapaxes(a, 1, 2)) * 1e-3))          # init-like scale
    d = rng.standard_normal((1500, s))
    e = rng.standard_normal((1500, s))
    check(d, e)
    check(d, e * 1e-9)                                                  # nearly diagonal


@pytest.mark.parametrize("gap", [1e-2, 1e-4, 1e-6, 1e-9, 1e-13])
def test_clusters_of_up_to_three(gap):
    """A pair and a triple of eigenvalues `gap` apart (relative to the spectrum): Gram-Schmidt inside the block keeps the
    vectors orthonormal whatever the gap."""
    rng = np.random.default_rng(7)
    vals = np.sort(rng.standard_normal((1500, 16)), 1)
    w = np.abs(vals).max(1)
    vals[:, 5] = vals[:, 4] + gap * w
    vals[:, 11] = vals[:, 10] + gap * w
    vals[:, 12] = vals[:, 11] + gap * w
    q, _ = np.linalg.qr(rng.standard_normal((1500, 16, 16)))
    a = (q * vals[:, None, :]) @ np.swapaxes(q, 1, 2)
    check(*tridiagonalize(0.5 * (a + np.swapaxes(a, 1, 2))))


def test_big_blocks_are_flagged_not_served():
    """More than INVIT_KEEP + 1 = 4 close eigenvalues: the routine says so (the kernel routes the pair to the QL-with-vectors
    kernel): a six-fold cluster, the identity (y = c x in the SPD backward), the zero matrix (y = x)."""
    rng = np.random.default_rng(9)
    vals = np.sort(rng.standard_normal((200, 16)), 1)
    for k in range(5, 10):
        vals[:, k] = vals[:, 4] + 1e-10 * (k - 4)
    q, _ = np.linalg.qr(rng.standard_normal((200, 16, 16)))
    a = (q * vals[:, None, :]) @ np.swapaxes(q, 1, 2)
    check(*tridiagonalize(0.5 * (a + np.swapaxes(a, 1, 2))), want_flagged=True)
    check(np.ones((20, 16)), np.zeros((20, 16)), want_flagged=True)
    check(np.zeros((20, 16)), np.zeros((20, 16)), want_flagged=True)


def test_wilkinson_like():
    d = np.tile(np.abs(np.arange(16) - 7.5), (4, 1))
    e = np.ones((4, 16))
    e[:, -1] = 0
    check(d, e)
