#!/usr/bin/env python3
"""SPD fixtures: tests/golden/spd_n{2,4,8,16}.npz  (python tools/make_golden_spd.py).

geoopt's SymmetricPositiveDefinite (the reference's `spd` model, sympa/embeddings.py:6,70-72,142) is absent from the
reference tree and not installed, so its outputs cannot be captured: SPD parity stays UNPINNED with respect to geoopt.
What these fixtures pin instead is the published formula itself, evaluated INDEPENDENTLY of every code path in this
repository (no Cholesky, no Householder, no QL, no torch): with mpmath at 50 digits,
    dist(x, y) = || log(x^-1/2 y x^-1/2) ||_F,     x^-1/2 and log through mp.eigsy,
plus, for the backward kernel, the Euclidean gradients of dist by 50-digit central differences along symmetric
directions.  Inputs and expected outputs only; nothing of geoopt or of the reference is stored.

    python tools/make_golden_spd.py --from-geoopt
closes the pin the day the dependency is provided: when `import geoopt` succeeds in the build container it evaluates geoopt's OWN
SymmetricPositiveDefinite (default metric, as sympa/embeddings.py:70 constructs it) -- dist, egrad2rgrad, retr, projx -- on the same
inputs and writes tests/golden/spd_geoopt_n{2,4,8,16}.npz (inputs + geoopt's outputs); tests/test_oracle_golden.py then compares
the oracle's spd_* restatements with those files (and with a live geoopt when it is importable) and SURVEY 8f-4 is pinned.  Without
geoopt the flag prints why it cannot run and exits 2; nothing is written."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpmath as mp  # noqa: E402

mp.mp.dps = 50


def sym(a):
    return 0.5 * (a + np.swapaxes(a, -1, -2))


def expm_sym(a):
    lam, v = np.linalg.eigh(a)
    return sym((v * np.exp(lam)[..., None, :]) @ np.swapaxes(v, -1, -2))


def mp_sym_funcm(a, f):
    lam, v = mp.eigsy(a)
    n = a.rows
    d = mp.zeros(n, n)
    for i in range(n):
        d[i, i] = f(lam[i])
    return v * d * v.T


def mp_dist(x, y):
    n = x.shape[0]
    mx = mp.matrix(n, n)
    my = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            mx[i, j] = mp.mpf(float(x[i, j]))
            my[i, j] = mp.mpf(float(y[i, j]))
    isq = mp_sym_funcm(mx, lambda t: 1 / mp.sqrt(t))
    inner = isq * my * isq
    inner = (inner + inner.T) / 2
    lam, _ = mp.eigsy(inner)
    return mp.sqrt(sum(mp.log(t) ** 2 for t in lam))


def cases(n, rng, b=10):
    out = {}
    eye = np.eye(n)
    out["init"] = (eye + sym(rng.uniform(-1e-3, 1e-3, (b, n, n))), eye + sym(rng.uniform(-1e-3, 1e-3, (b, n, n))))
    for s in (0.1, 0.5, 1.5):
        out[f"s{s}"] = (expm_sym(sym(rng.normal(size=(b, n, n)) * s)), expm_sym(sym(rng.normal(size=(b, n, n)) * s)))
    x = expm_sym(sym(rng.normal(size=(b, n, n)) * 0.4))
    out["same"] = (x, x.copy())
    dx, dy = np.exp(rng.normal(size=(b, n)) * 0.7), np.exp(rng.normal(size=(b, n)) * 0.7)
    out["diag"] = (np.einsum("bi,ij->bij", dx, eye), np.einsum("bi,ij->bij", dy, eye))
    # ill-conditioned x (cond ~1e6) against a well-conditioned y
    lam = np.exp(np.linspace(-7, 7, n))[None] * np.exp(rng.normal(size=(b, n)) * 0.1)
    q, _ = np.linalg.qr(rng.normal(size=(b, n, n)))
    out["cond1e6"] = (sym((q * lam[:, None, :]) @ np.swapaxes(q, -1, -2)), expm_sym(sym(rng.normal(size=(b, n, n)) * 0.3)))
    return out


def main():
    for n in (2, 4, 8, 16):
        rng = np.random.default_rng(1000 + n)
        blob = {"case_names": []}
        for name, (x, y) in cases(n, rng, b=10 if n <= 8 else 6).items():
            blob["case_names"].append(name)
            blob[f"{name}__x"], blob[f"{name}__y"] = x, y
            blob[f"{name}__dist_exact50"] = np.array([float(mp_dist(x[k], y[k])) for k in range(x.shape[0])])
        # directional derivatives of dist for the backward kernel: d/dt dist(x + t S, y) and d/dt dist(x, y + t S)
        # along random symmetric S, by 50-digit central differences (h = 1e-20: truncation error ~1e-40)
        x, y = blob["s0.5__x"][:4], blob["s0.5__y"][:4]
        dirs = sym(rng.normal(size=(4, 3, n, n)))
        h = 1e-20
        ddx = np.zeros((4, 3))
        ddy = np.zeros((4, 3))
        for k in range(4):
            for t in range(3):
                s = dirs[k, t]

                def shifted(base, sign):
                    m = mp.matrix(n, n)
                    for i in range(n):
                        for j in range(n):
                            m[i, j] = mp.mpf(float(base[i, j])) + sign * mp.mpf(h) * mp.mpf(float(s[i, j]))
                    return m

                def dist_m(mx, my):
                    isq = mp_sym_funcm(mx, lambda u: 1 / mp.sqrt(u))
                    inner = isq * my * isq
                    inner = (inner + inner.T) / 2
                    lam, _ = mp.eigsy(inner)
                    return mp.sqrt(sum(mp.log(u) ** 2 for u in lam))
                my = shifted(y[k], 0)
                mx = shifted(x[k], 0)
                ddx[k, t] = float((dist_m(shifted(x[k], 1), my) - dist_m(shifted(x[k], -1), my)) / (2 * mp.mpf(h)))
                ddy[k, t] = float((dist_m(mx, shifted(y[k], 1)) - dist_m(mx, shifted(y[k], -1))) / (2 * mp.mpf(h)))
        blob["grad__x"], blob["grad__y"], blob["grad__dirs"] = x, y, dirs
        blob["grad__ddx_exact50"], blob["grad__ddy_exact50"] = ddx, ddy
        blob["case_names"] = np.array(blob["case_names"])
        path = os.path.join(ROOT, "tests", "golden", f"spd_n{n}.npz")
        np.savez_compressed(path, **blob)
        print(path, os.path.getsize(path), "bytes")


def from_geoopt():
    """geoopt's own outputs on the fixtures' inputs (only when geoopt is importable: it is absent from /root/reference and from
    this image)."""
    try:
        import geoopt
        import torch
    except ImportError as e:
        print(f"--from-geoopt: geoopt is not importable here ({e}); spd parity stays UNPINNED, nothing written")
        return 2
    torch.set_default_dtype(torch.float64)
    man = geoopt.manifolds.SymmetricPositiveDefinite()            # sympa/embeddings.py:70: the default (affine-invariant) metric
    for n in (2, 4, 8, 16):
        rng = np.random.default_rng(1000 + n)
        blob = {"case_names": [], "geoopt_version": np.array(getattr(geoopt, "__version__", "unknown"))}
        for name, (x, y) in cases(n, rng, b=10 if n <= 8 else 6).items():
            tx, ty = torch.from_numpy(x), torch.from_numpy(y)
            u = torch.from_numpy(sym(rng.normal(size=x.shape)) * 0.1)
            blob["case_names"].append(name)
            blob[f"{name}__x"], blob[f"{name}__y"], blob[f"{name}__u"] = x, y, u.numpy()
            blob[f"{name}__dist"] = man.dist(tx, ty).numpy()
            blob[f"{name}__egrad2rgrad"] = man.egrad2rgrad(tx, u).numpy()
            blob[f"{name}__retr"] = man.retr(tx, 0.05 * man.egrad2rgrad(tx, u)).numpy()
            blob[f"{name}__projx"] = man.projx(tx + u).numpy()
        blob["case_names"] = np.array(blob["case_names"])
        path = os.path.join(ROOT, "tests", "golden", f"spd_geoopt_n{n}.npz")
        np.savez_compressed(path, **blob)
        print(path, os.path.getsize(path), "bytes")
    return 0


if __name__ == "__main__":
    if "--from-geoopt" in sys.argv[1:]:
        sys.exit(from_geoopt())
    main()
