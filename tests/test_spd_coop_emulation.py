"""CPU emulation of the sixteen-lanes-per-pair SPD kernel (sympa_amd/csrc/spd_coop.hpp): the same sequence of
row-per-lane operations (right-looking Cholesky, two right solves with a transpose in between, Householder with the
reflector taken from the COLUMN held across the lanes, lockstep QL), written with numpy arrays whose first axis is the
lane.  It pins the numerical design choices of the kernel on the CPU; the kernel itself is checked on the GPU against
the oracle and against the one-lane-per-pair kernel (tests/test_spd.py)."""
import numpy as np
import pytest
import torch

from oracle import siegel_oracle as so
from tests.helpers import spd_points

N = 16


def coop_tridiagonal(X, Y, reflector="column"):
    lane = np.arange(N)
    x = X.copy()                       # x[i, :] = registers of lane i
    y = Y - X
    rd = np.zeros(N)
    for j in range(N):                 # Cholesky, right-looking: column j scaled, then rank-1 update of the rest
        rr = 1.0 / np.sqrt(x[j, j])
        rd[j] = rr
        x[:, j] *= rr
        for k in range(j + 1, N):
            x[:, k] -= x[k, j] * x[:, j]

    def solve_right_lt(a):             # a <- a L^-T, every lane on its own row
        for j in range(N):
            for k in range(j):
                a[:, j] -= x[j, k] * a[:, k]
            a[:, j] *= rd[j]

    solve_right_lt(y)
    m = y.T.copy()                     # transpose through the LDS
    solve_right_lt(m)
    d, e2 = np.zeros(N), np.zeros(N)
    for k in range(N - 2):
        col = m[:, k].copy()           # lane i: its element of column k
        row = m[k, :].copy()           # lane k: its whole row (the variant the kernel does NOT use)
        x0, dk = col[k + 1], col[k]
        s2 = float((np.where(lane > k + 1, col, 0.0) ** 2).sum())
        n2 = x0 * x0 + s2
        d[k], e2[k] = dk, n2
        v0 = x0 + np.copysign(np.sqrt(n2), x0)
        den = v0 * v0 + s2
        beta = 2.0 / den if den > 0 else 0.0
        vi = np.where(lane <= k, 0.0, np.where(lane == k + 1, v0, col))
        vb = vi if reflector == "column" else np.where(lane <= k, 0.0, np.where(lane == k + 1, v0, row))
        p = m[:, k + 1:] @ vb[k + 1:]
        p = np.where(lane <= k, 0.0, beta * p)
        kk = 0.5 * beta * float((vi * p).sum())
        q = p - kk * vi
        for j in range(k + 1, N):
            m[:, j] -= vb[j] * q
            m[:, j] -= q[j] * vi
    d[N - 2], d[N - 1], e2[N - 2] = m[N - 2, N - 2], m[N - 1, N - 1], m[N - 2, N - 1] ** 2
    return d, e2


def lockstep_ql(D, E2):
    """tridiag_ql_lockstep (csrc/siegel_math.hpp) over a 'wave' of tridiagonals: D, E2 are [lanes, N]."""
    d, e2 = D.copy(), E2.copy()
    lanes = np.arange(len(d))
    tiny = 1e-300

    def negligible(i):
        return e2[:, i] <= 1.3e-32 * np.abs(d[:, i] * d[:, i + 1]) + 1e-290

    iterations = 0
    for L in range(N - 1):
        for _ in range(60):
            conv = negligible(L)
            if conv.all():
                break
            iterations += 1
            pos = np.full(len(d), L)
            idle = conv.copy()
            e2[conv, L] = 0.0
            for a in (1, 2):
                if L + a <= N - 2:
                    pos = np.where(idle, L + a, pos)
                    idle = idle & negligible(L + a)
                    e2[idle, L + a] = 0.0
            dl, dl1 = d[lanes, pos], d[lanes, pos + 1]
            el = np.where(idle, 1.0, e2[lanes, pos])
            rte = np.sqrt(el)
            sg = 0.5 * (dl1 - dl) / rte
            sigma = np.where(idle, dl, dl - rte / (sg + np.copysign(np.sqrt(sg * sg + 1.0), sg)))
            c, sn = np.ones(len(d)), np.zeros(len(d))
            gamma = d[:, N - 1] - sigma
            p = gamma * gamma
            for i in range(N - 2, L - 1, -1):
                bb = e2[:, i]
                r = p + bb
                if i != N - 2:
                    e2[:, i + 1] = sn * r
                oldc = c
                rs = np.maximum(r, tiny)
                c = (p + (rs - r)) / rs
                sn = bb / rs
                oldgam = gamma
                alpha = d[:, i].copy()
                gamma = c * (alpha - sigma) - sn * oldgam
                d[:, i + 1] = oldgam + (alpha - gamma)
                with np.errstate(divide="ignore", invalid="ignore"):
                    p = np.where(c != 0.0, gamma * gamma / c, oldc * bb)
            e2[:, L] = sn * p
            d[:, L] = sigma + gamma
        else:
            raise AssertionError("QL did not converge")
    return d, iterations


def coop_distance(x, y, reflector="column"):
    tri = [coop_tridiagonal(x[i], y[i], reflector) for i in range(len(x))]
    ev, iters = lockstep_ql(np.stack([t[0] for t in tri]), np.stack([t[1] for t in tri]))
    return np.sqrt((np.log1p(ev) ** 2).sum(-1)), iters


@pytest.mark.parametrize("s,tol", [(1e-3, 1e-12), (0.3, 1e-12), (0.6, 1e-10)])
def test_lane_algorithm_matches_the_oracle(s, tol):
    g = torch.Generator().manual_seed(77)
    x, y = spd_points(256, N, s, g), spd_points(256, N, s, g)
    want = so.spd_dist(x, y).numpy()
    got, iters = coop_distance(x.numpy(), y.numpy())
    assert np.max(np.abs(got - want) / want) < tol
    assert iters < 60          # ~45 sweeps for the slowest of 256 lanes, against 40 N for the bound of dsterf


def test_identical_points_and_exact_zero():
    g = torch.Generator().manual_seed(78)
    x = spd_points(64, N, 0.3, g).numpy()
    got, _ = coop_distance(x, x)
    assert np.all(got == 0.0)


def test_reflector_must_come_from_one_source():
    """Mixing lane k's row with the lanes' own column elements loses accuracy when the eliminated column is small
    (DESIGN.md section 10): the row variant is measurably worse on the same inputs."""
    g = torch.Generator().manual_seed(1616)
    x, y = spd_points(1000, N, 0.3, g), spd_points(1000, N, 0.3, g)
    want = so.spd_dist(x, y).numpy()
    col, _ = coop_distance(x.numpy(), y.numpy(), "column")
    row, _ = coop_distance(x.numpy(), y.numpy(), "row")
    err_col = np.max(np.abs(col - want) / want)
    err_row = np.max(np.abs(row - want) / want)
    assert err_col < 2e-14
    assert err_row > 5 * err_col
