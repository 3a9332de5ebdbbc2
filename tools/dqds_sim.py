#!/usr/bin/env python3
"""CPU simulation (numpy, 64 lanes = one wave per block) of the eigenvalue stage of the spd / Siegel n >= 5 kernels on the
tridiagonal forms of the bench tables: the lockstep PWK QL the kernels run today against lockstep dqds variants on the
shifted positive-definite form (round-3 review, item 1).  Counts ELEMENT-SWEEPS per wave (one element of one sweep, executed
by the whole wave) and weights them with the per-element instruction cost of each inner loop; checks eigenvalues against
numpy.linalg.eigvalsh.

    python tools/dqds_sim.py [spd|siegel] [n] [waves]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def tridiagonalize(a):
    """Householder tridiagonalisation, batch [B, n, n] symmetric -> d [B, n], e [B, n-1] (off-diagonal, signed)."""
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
        kk = 0.5 * beta * (v * p).sum(1)
        p -= kk[:, None] * v
        sub -= v[:, :, None] * p[:, None, :] + p[:, :, None] * v[:, None, :]
        a[:, k + 1, k] = -np.copysign(nx, x[:, 0])
        a[:, k, k + 1] = a[:, k + 1, k]
        a[:, k + 2:, k] = 0
        a[:, k, k + 2:] = 0
    d = np.einsum("bii->bi", a).copy()
    e = np.stack([a[:, i + 1, i] for i in range(n - 1)], 1)
    return d, e


def spd_forms(n, pairs, seed=42):
    from sympa_amd import data
    nodes = 100000 if n == 16 else 5000
    tab = data.spd_table(nodes, n, seed=seed).numpy()
    pr = data.sample_pairs(nodes, pairs, 0, seed).numpy()
    x, y = tab[pr[:, 0]], tab[pr[:, 1]]
    l = np.linalg.cholesky(x)
    li = np.linalg.inv(l)
    m = li @ (y - x) @ np.swapaxes(li, -1, -2)
    m = 0.5 * (m + np.swapaxes(m, -1, -2))
    return tridiagonalize(m), np.linalg.eigvalsh(m)


def siegel_forms(n, pairs, seed=42):
    from sympa_amd import data
    nodes = 45500 if n == 8 else 5041
    tab = data.trained_like_table(nodes, n, model="upper", seed=seed).numpy()
    pr = data.sample_pairs(nodes, pairs, 0, seed).numpy()
    z1, z2 = tab[pr[:, 0]], tab[pr[:, 1]]
    l1, l2 = np.linalg.cholesky(z1[:, 1]), np.linalg.cholesky(z2[:, 1])
    dz = (z2[:, 0] - z1[:, 0]) + 1j * (z2[:, 1] - z1[:, 1])
    e = np.linalg.inv(l1) @ dz @ np.swapaxes(np.linalg.inv(l2), -1, -2)
    h = np.swapaxes(e.conj(), -1, -2) @ e
    # Hermitian -> real symmetric tridiagonal has the same eigenvalues as the complex Householder form; use the real 2n embedding's
    # spectrum for checking and a unitary reduction by numpy for the form
    w = np.linalg.eigvalsh(h)
    # complex Householder tridiagonalisation
    a = h.copy()
    b = a.shape[0]
    for k in range(n - 2):
        x = a[:, k + 1:, k].copy()
        nx = np.linalg.norm(x, axis=1)
        ph = np.where(np.abs(x[:, 0]) > 0, x[:, 0] / np.where(np.abs(x[:, 0]) > 0, np.abs(x[:, 0]), 1), 1.0)
        v = x.copy()
        v[:, 0] += ph * nx
        den = (np.abs(v) ** 2).sum(1)
        beta = np.where(den > 0, 2.0 / np.where(den > 0, den, 1.0), 0.0)
        sub = a[:, k + 1:, k + 1:]
        p = beta[:, None] * np.einsum("bij,bj->bi", sub, v)
        kk = 0.5 * beta * (v.conj() * p).sum(1)
        p -= kk[:, None] * v
        sub -= v[:, :, None] * p.conj()[:, None, :] + p[:, :, None] * v.conj()[:, None, :]
        a[:, k + 1, k] = -ph * nx
        a[:, k, k + 1] = np.conj(a[:, k + 1, k])
        a[:, k + 2:, k] = 0
        a[:, k, k + 2:] = 0
    d = np.einsum("bii->bi", a).real.copy()
    e = np.stack([np.abs(a[:, i + 1, i]) for i in range(n - 1)], 1)
    return (d, e), w


TINY = 1e-290
VARIANTS = ("rut:1.0", "rut:0.95", "rut:0.8", "rut:0.5")


def negligible(e2, da, db):
    return ~(e2 > 1.3e-32 * np.abs(da * db) + 1e-290)


def ql_lockstep(d, e, reverse=True):
    """tridiag_ql_lockstep of siegel_math.hpp on one wave [64, n] (PWK, squared off-diagonals).  Returns eigenvalues and
    the element-sweeps the wave executed."""
    d = d.copy()
    e2 = np.concatenate([e * e, np.zeros((d.shape[0], 1))], 1)
    if reverse:
        d = d[:, ::-1].copy()
        e2 = np.concatenate([e2[:, :-1][:, ::-1], np.zeros((d.shape[0], 1))], 1)
    n = d.shape[1]
    work = 0
    for L in range(n - 1):
        for it in range(60):
            conv = negligible(e2[:, L], d[:, L], d[:, L + 1])
            if conv.all():
                break
            dl, dl1, el = d[:, L].copy(), d[:, L + 1].copy(), e2[:, L].copy()
            idle = conv.copy()
            e2[:, L] = np.where(conv, 0.0, e2[:, L])
            for look in (1, 2):
                if L + look <= n - 2:
                    c = negligible(e2[:, L + look], d[:, L + look], d[:, L + look + 1])
                    dl = np.where(idle, d[:, L + look], dl)
                    dl1 = np.where(idle, d[:, L + look + 1], dl1)
                    el = np.where(idle, e2[:, L + look], el)
                    idle = idle & c
                    e2[:, L + look] = np.where(idle, 0.0, e2[:, L + look])
            el = np.where(idle, 1.0, el)
            rte = np.sqrt(el)
            sg = 0.5 * (dl1 - dl) / rte
            rr = np.sqrt(sg * sg + 1.0)
            sigma = dl - rte / (sg + np.copysign(rr, sg))
            sigma = np.where(idle, dl, sigma)
            c = np.ones_like(sigma)
            sn = np.zeros_like(sigma)
            gamma = d[:, n - 1] - sigma
            p = gamma * gamma
            for i in range(n - 2, L - 1, -1):
                bb = e2[:, i]
                r = p + bb
                if i != n - 2:
                    e2[:, i + 1] = sn * r
                oldc = c
                rs = np.maximum(r, TINY)
                c = (p + (rs - r)) / rs
                sn = bb / rs
                oldgam = gamma
                alpha = d[:, i]
                gamma = c * (alpha - sigma) - sn * oldgam
                d[:, i + 1] = oldgam + (alpha - gamma)
                with np.errstate(divide="ignore", invalid="ignore"):
                    p = np.where(c != 0.0, gamma * gamma / np.where(c != 0, c, 1.0), oldc * bb)
                work += 1
            e2[:, L] = sn * p
            d[:, L] = sigma + gamma
    return d, work


def dqds_lockstep(d, e, variant="safe", look=2, verbose=False):
    """Lockstep dqds on one wave.  The tridiagonal (d, e) is shifted by its Gershgorin lower bound g (T - g I is positive
    semidefinite), factored T - g I = L D L^T -> qd arrays (q, ee), and every lane runs dqds sweeps over the leading block
    [0, last]; `last` goes n-1, n-2, ... when ALL lanes have a negligible ee[last - 1].  A lane that is through at `last`
    records lambda = sigma + q[last], marks the position (q = BIG, decoupled) and goes on with its own bottom position
    (up to `look` positions ahead of the wave's).  Shifts (per lane):
      safe:   s = max(0, qb - sqrt(qb * eb) ...) never above the smallest eigenvalue (no failures)
      aggr:   Rutishauser-like aggressive estimate, a failed sweep (negative d) is rolled back by the lane and retried
              with a quarter of the shift (costs the lane the sweep, not the wave)
    Returns eigenvalues (unsorted) and element-sweeps."""
    B, n = d.shape
    off = np.zeros((B, n))
    off[:, :-1] += np.abs(e)
    off[:, 1:] += np.abs(e)
    g = (d - off).min(1)
    g = g - 1e-3 * np.abs(g) - 1e-300           # strictly below the spectrum
    # L D L^T of T - g: q_0 = d_0 - g; ee_i = e_i^2 / q_i; q_{i+1} = d_{i+1} - g - ee_i
    q = np.zeros((B, n))
    ee = np.zeros((B, n))
    q[:, 0] = d[:, 0] - g
    for i in range(n - 1):
        ee[:, i] = e[:, i] ** 2 / q[:, i]
        q[:, i + 1] = d[:, i + 1] - g - ee[:, i]
    assert (q > 0).all(), q.min()
    sigma = g.copy()
    lam = np.full((B, n), np.nan)
    gmax = (d + off).max(1)
    BIG = (4.0 * (gmax - g) + 1e-300)[:, None] * np.ones((1, n))      # a decoupled position: large, never the minimum
    bottom = np.full(B, n - 1)                 # per-lane bottom position (>= wave's `last` - look)
    work = 0
    fails = 0
    prev_dmin = np.full(B, np.inf)
    shrink = np.ones(B)
    late_ok = np.zeros(B, bool)
    late_shift = np.zeros(B)
    theta = float(variant.split(":")[1]) if ":" in variant else 0.9
    for last in range(n - 1, 0, -1):
        for it in range(80):
            # deflation bookkeeping for positions last, last-1, ..., last-look
            for pos in range(last, max(last - look, 0) - 1, -1):
                at = bottom == pos
                if pos == 0:
                    done = at
                else:
                    done = at & ~(ee[:, pos - 1] > 1e-30 * (np.abs(sigma) + q[:, pos]) * 1e-2 + 0)
                    # negligible: ee tiny relative to the eigenvalue scale (absolute criterion on sigma + q)
                    done = at & (ee[:, pos - 1] <= 2.5e-17 * np.abs(sigma + q[:, pos]) + 1e-290)
                if done.any():
                    lam[done, pos] = sigma[done] + q[done, pos]
                    q[done, pos] = BIG[done, pos]
                    if pos > 0:
                        ee[done, pos - 1] = 0.0
                    bottom[done] = pos - 1
                    prev_dmin[done] = np.inf
                    shrink[done] = 1.0
                    late_ok[done] = False
            if (bottom < last).all():
                break
            # shift per lane from its bottom 2 x 2 (positions b-1, b of the active block)
            b = np.clip(bottom, 0, n - 1)
            idx = np.arange(B)
            qb = q[idx, b]
            live = bottom >= 1
            eb = np.where(live, ee[idx, np.clip(b - 1, 0, n - 1)], 0.0)
            qa = np.where(live, q[idx, np.clip(b - 1, 0, n - 1)], 1.0)
            if variant == "safe":
                # smallest eigenvalue of the trailing 2x2 of B^T B is NOT a lower bound of the block's; the guaranteed one used
                # here: Gershgorin on the last row of B^T B: alpha_b - beta_{b-1} = (qb + eb) - sqrt(qa eb) ... can be < 0;
                # combine with the dmin bound of the previous sweep
                s = np.maximum(0.0, qb - np.sqrt(np.maximum(qa * eb, 0.0)))
                s = np.minimum(s, np.where(np.isfinite(prev_dmin), prev_dmin, s))
                s = np.where(live, s, 0.0)
            else:
                # upper bounds of the smallest eigenvalue of the active block: the smaller eigenvalue of its trailing 2 x 2 (a
                # principal submatrix: interlacing) and dmin of the last successful sweep.  A shift must stay BELOW lambda_min:
                #   rut:  after a LATE failure (only the last d negative) s_fail + d_last is a guaranteed lower bound
                #         (Rutishauser); otherwise theta x the upper bound, theta shrinking with every early failure
                a11 = qa + np.where(bottom >= 2, ee[idx, np.clip(b - 2, 0, n - 1)], 0.0)
                a22 = qb + eb
                a12sq = qa * eb
                tr, det = a11 + a22, a11 * a22 - a12sq
                disc = np.sqrt(np.maximum(0.25 * tr * tr - det, 0.0))
                small = np.maximum(det, 0.0) / (0.5 * tr + disc)            # smaller root, stable form
                ub = np.minimum(small, np.where(np.isfinite(prev_dmin), prev_dmin, small))
                s = np.where(late_ok, late_shift, ub * theta * shrink)
                s = np.where(live, np.maximum(s, 0.0), 0.0)
                s = np.where(shrink < 0.01, 0.0, s)         # repeated early failures: a zero-shift (dqd) sweep cannot fail
            # one dqds sweep over [0, last] (positions beyond a lane's bottom are decoupled BIGs: pass through)
            qq = q.copy()
            en = ee.copy()
            dd = q[:, 0] - s
            dmin = dd.copy()
            dlast = dd.copy()
            bad = dd < 0
            early = bad.copy()
            for i in range(last):
                qq[:, i] = dd + ee[:, i]
                t = q[:, i + 1] / np.maximum(qq[:, i], 1e-150)
                en[:, i] = ee[:, i] * t
                dd = dd * t - s
                active = (i + 1) <= bottom
                bad |= active & ~(dd >= 0)
                early |= ((i + 1) < bottom) & ~(dd >= 0)
                dlast = np.where((i + 1) == bottom, dd, dlast) if i > 0 or True else dlast
                dmin = np.where(active, np.minimum(dmin, dd), dmin)
                work += 1
            qq[:, last] = dd
            ok = ~bad
            fails += int(bad.sum())
            # successful lanes take the new arrays; failed lanes keep the old ones and shrink their shift
            q = np.where(ok[:, None], qq, q)
            ee = np.where(ok[:, None], en, ee)
            for pos in range(n):               # (simulation only: decoupled positions are reset every sweep)
                beyond = pos > bottom
                q[:, pos] = np.where(beyond, BIG[:, pos], q[:, pos])
                if pos > 0:
                    ee[:, pos - 1] = np.where(beyond, 0.0, ee[:, pos - 1])
            sigma = np.where(ok, sigma + s, sigma)
            prev_dmin = np.where(ok, dmin, prev_dmin)
            late = bad & ~early & np.isfinite(dlast)          # only the bottom d went negative
            late_shift = np.where(late, s + dlast, 0.0)
            late_ok = late & (late_shift > 0)
            shrink = np.where(ok, 1.0, np.where(late, shrink, shrink * 0.25))
        else:
            raise RuntimeError(f"no convergence at last={last}")
    # position 0
    rem = bottom == 0
    lam[rem, 0] = sigma[rem] + q[rem, 0]
    return lam, work, fails


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "spd"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    waves = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    (d, e), want = (spd_forms if kind == "spd" else siegel_forms)(n, 64 * waves)
    scale = np.abs(want).max(1)
    tot = dict({"ql": 0}, **{v: 0 for v in VARIANTS})
    err = dict({"ql": 0.0}, **{v: 0.0 for v in VARIANTS})
    fails = {v: 0 for v in VARIANTS}
    for w in range(waves):
        sl = slice(64 * w, 64 * (w + 1))
        got, work = ql_lockstep(d[sl], e[sl])
        tot["ql"] += work
        err["ql"] = max(err["ql"], float((np.abs(np.sort(got, 1) - want[sl]).max(1) / scale[sl]).max()))
        for variant in VARIANTS:
            lam, work, f = dqds_lockstep(d[sl], e[sl], variant)
            tot[variant] += work
            fails[variant] += f
            err[variant] = max(err[variant], float((np.abs(np.sort(lam, 1) - want[sl]).max(1) / scale[sl]).max()))
    print(f"{kind} n={n}: {waves} waves")
    for k, cost in [("ql", 30)] + [(v, 16) for v in VARIANTS]:
        print(f"  {k:5s} element-sweeps per wave {tot[k] / waves:8.1f}   x {cost} instr = {tot[k] / waves * cost:9.0f}   "
              f"max |err| / |lambda|_max {err[k]:.2e}   failed lane-sweeps {fails.get(k, 0)}")


if __name__ == "__main__":
    main()
