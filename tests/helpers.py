"""Shared helpers for the tests: golden loading, synthetic points, hostsim binding."""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
METRICS = ["riem", "fone", "finf", "fmin", "wsum"]
MODELS = ["upper", "bounded"]


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float64)


def rel_err(got, want, atol=1e-12):
    """max over the batch of max(|got-want| - atol, 0) / |want|: relative error with an absolute
    floor for d ~ 0 (the reference's own d(x,x) is ~1e-15, not 0)."""
    got, want = torch.as_tensor(got), torch.as_tensor(want)
    excess = ((got - want).abs() - atol).clamp_min(0.0)
    return (excess / want.abs().clamp_min(1e-300)).max().item()


def sym(x):
    return 0.5 * (x + x.transpose(-1, -2))


def upper_points(b, n, s, g):
    x = sym(torch.randn(b, n, n, generator=g, dtype=torch.float64) * s)
    y = torch.matrix_exp(sym(torch.randn(b, n, n, generator=g, dtype=torch.float64) * s))
    return torch.stack((x, sym(y)), 1)


def to_bounded(z):
    """Cayley image, exactly symmetric (what a projected bounded table row looks like)."""
    zc = torch.complex(z[:, 0], z[:, 1])
    eye = torch.eye(z.shape[-1], dtype=zc.dtype)
    w = (zc - 1j * eye) @ torch.linalg.inv(zc + 1j * eye)
    w = 0.5 * (w + w.transpose(-1, -2))
    return torch.stack((w.real, w.imag), 1)


def points(model, b, n, s, g):
    z = upper_points(b, n, s, g)
    return to_bounded(z) if model == "bounded" else z


_hostsim = None


def hostsim():
    """CPU build of the kernel arithmetic (tests/hostsim), built on demand with g++."""
    global _hostsim
    if _hostsim is None:
        d = os.path.join(ROOT, "tests", "hostsim")
        so = os.path.join(d, "libsympa_hostsim.so")
        srcs = [os.path.join(d, "hostsim.cpp"), os.path.join(ROOT, "sympa_amd", "csrc", "siegel_math.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "siegel_math_bwd.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "siegel_math_bwd_split.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "siegel_table_math.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "siegel_math_generic.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "spd_math.hpp"),
                os.path.join(ROOT, "sympa_amd", "csrc", "spd_math_bwd.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, srcs[0]], cwd=d)
        _hostsim = ctypes.CDLL(so)
    return _hostsim


def hostsim_dist(z1, z2, model, metric, weights=None, eps=1e-5, generic=False):
    lib = hostsim()
    P = ctypes.c_void_p
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    z2 = np.ascontiguousarray(z2, dtype=np.float64)
    b, _, n, _ = z1.shape
    out = np.zeros(b)
    vvd = np.zeros((b, n))
    st = ctypes.c_int32(0)
    w = np.ascontiguousarray(np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1))
    fn = lib.sympa_hostsim_dist_generic if generic else lib.sympa_hostsim_dist
    rc = fn(P(z1.ctypes.data), P(z2.ctypes.data), ctypes.c_int64(b), n,
                                MODELS.index(model), METRICS.index(metric), P(w.ctypes.data),
                                ctypes.c_double(eps), P(out.ctypes.data), P(vvd.ctypes.data), ctypes.byref(st))
    assert rc == 0, rc
    return out, vvd, st.value


def hostsim_dist_packed(z1, z2, model, metric, weights=None, eps=1e-5, diff=False):
    """The all-pairs kernel's per-pair arithmetic: packed points (inverted factors), E = A1 (Z2 - Z1) A2^T.  diff=True: the indexed
    packed forward's form (csrc/siegel_packed_kernel.hpp): the first point's triangles subtracted from the second's packed row in
    place, e_from_packed<DIFF>, distance_from_h."""
    lib = hostsim()
    P = ctypes.c_void_p
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    z2 = np.ascontiguousarray(z2, dtype=np.float64)
    b, _, n, _ = z1.shape
    out = np.zeros(b)
    st = ctypes.c_int32(0)
    w = np.ascontiguousarray(np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1))
    rc = lib.sympa_hostsim_dist_packed(P(z1.ctypes.data), P(z2.ctypes.data), ctypes.c_int64(b), n, MODELS.index(model),
                                       METRICS.index(metric) + (16 if diff else 0), P(w.ctypes.data), ctypes.c_double(eps),
                                       P(out.ctypes.data), ctypes.byref(st))
    assert rc == 0, rc
    return out, st.value


def hostsim_dist_bwd(z1, z2, go, model, metric, weights=None, eps=1e-5):
    lib = hostsim()
    P = ctypes.c_void_p
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    z2 = np.ascontiguousarray(z2, dtype=np.float64)
    go = np.ascontiguousarray(go, dtype=np.float64)
    b, _, n, _ = z1.shape
    out = np.zeros(b)
    g1 = np.zeros_like(z1)
    g2 = np.zeros_like(z2)
    gw = np.zeros(n)
    st = ctypes.c_int32(0)
    w = np.ascontiguousarray(np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1))
    rc = lib.sympa_hostsim_dist_bwd(P(z1.ctypes.data), P(z2.ctypes.data), P(go.ctypes.data), ctypes.c_int64(b), n,
                                    MODELS.index(model), METRICS.index(metric), P(w.ctypes.data), ctypes.c_double(eps),
                                    P(out.ctypes.data), P(g1.ctypes.data), P(g2.ctypes.data), P(gw.ctypes.data),
                                    ctypes.byref(st))
    assert rc == 0, rc
    return out, g1, g2, gw, st.value


def hostsim_dist_bwd_split(z1, z2, go, model, metric, weights=None, eps=1e-5):
    """The two-stage adjoint of dims 5..8 (siegel_math_bwd_split.hpp: stage 1 -> pack scaled by go -> stage 2) on the CPU build;
    same returns as hostsim_dist_bwd."""
    lib = hostsim()
    P = ctypes.c_void_p
    z1 = np.ascontiguousarray(z1, dtype=np.float64)
    z2 = np.ascontiguousarray(z2, dtype=np.float64)
    go = np.ascontiguousarray(go, dtype=np.float64)
    b, _, n, _ = z1.shape
    out = np.zeros(b)
    g1 = np.zeros_like(z1)
    g2 = np.zeros_like(z2)
    gw = np.zeros(n)
    st = ctypes.c_int32(0)
    w = np.ascontiguousarray(np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1))
    rc = lib.sympa_hostsim_dist_bwd_split(P(z1.ctypes.data), P(z2.ctypes.data), P(go.ctypes.data), ctypes.c_int64(b), n,
                                          MODELS.index(model), METRICS.index(metric), P(w.ctypes.data), ctypes.c_double(eps),
                                          P(out.ctypes.data), P(g1.ctypes.data), P(g2.ctypes.data), P(gw.ctypes.data),
                                          ctypes.byref(st))
    assert rc == 0, rc
    return out, g1, g2, gw, st.value


def hostsim_table(op, model, z, g=None, lr=0.0, wd=0.0, eps=1e-5):
    """op: 'projx' | 'rsgd' | 'egrad2rgrad' -> (rows, projected_count)"""
    lib = hostsim()
    P = ctypes.c_void_p
    z = np.ascontiguousarray(z, dtype=np.float64)
    b, _, n, _ = z.shape
    out = np.zeros_like(z)
    gp = None
    if g is not None:
        g = np.ascontiguousarray(g, dtype=np.float64)
        gp = P(g.ctypes.data)
    moved = ctypes.c_int32(0)
    st = lib.sympa_hostsim_table({"projx": 0, "rsgd": 1, "egrad2rgrad": 2}[op], MODELS.index(model), n, P(z.ctypes.data),
                                 gp, P(out.ctypes.data), ctypes.c_int64(b), ctypes.c_double(lr), ctypes.c_double(wd),
                                 ctypes.c_double(eps), ctypes.byref(moved))
    assert st == 0, st
    return out, moved.value


def spd_points(b, n, s, g):
    a = sym(torch.randn(b, n, n, generator=g, dtype=torch.float64) * s)
    return sym(torch.matrix_exp(a))


def hostsim_spd_dist(x, y):
    lib = hostsim()
    P = ctypes.c_void_p
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(x.shape[0])
    st = ctypes.c_int32(0)
    rc = lib.sympa_hostsim_spd_dist(P(x.ctypes.data), P(y.ctypes.data), ctypes.c_int64(x.shape[0]), x.shape[1],
                                    P(out.ctypes.data), ctypes.byref(st))
    assert rc == 0
    return out, st.value


def hostsim_spd_bwd(x, y):
    """(dist, d dist / dx, d dist / dy) from the g++ build of spd_math_bwd.hpp"""
    lib = hostsim()
    P = ctypes.c_void_p
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(x.shape[0])
    gx, gy = np.zeros_like(x), np.zeros_like(y)
    st = ctypes.c_int32(0)
    rc = lib.sympa_hostsim_spd_bwd(P(x.ctypes.data), P(y.ctypes.data), ctypes.c_int64(x.shape[0]), x.shape[1],
                                   P(out.ctypes.data), P(gx.ctypes.data), P(gy.ctypes.data), ctypes.byref(st))
    assert rc == 0
    return out, gx, gy, st.value


def hostsim_spd_table(op, x, g=None, lr=0.0, wd=0.0):
    """op: 'projx' | 'rsgd' | 'egrad2rgrad' -> (rows, moved)"""
    lib = hostsim()
    P = ctypes.c_void_p
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    gp = None
    if g is not None:
        g = np.ascontiguousarray(g, dtype=np.float64)
        gp = P(g.ctypes.data)
    moved = ctypes.c_int32(0)
    st = lib.sympa_hostsim_spd_table({"projx": 0, "rsgd": 1, "egrad2rgrad": 2}[op], x.shape[1], P(x.ctypes.data), gp,
                                     P(out.ctypes.data), ctypes.c_int64(x.shape[0]), ctypes.c_double(lr),
                                     ctypes.c_double(wd), ctypes.byref(moved))
    assert st == 0, st
    return out, moved.value


def hostsim_tridiag_invit(d, e):
    """Eigen-decomposition of symmetric tridiagonals the way the three-kernel SPD backward does it (g++ build of
    tridiag_invit.hpp behind the lockstep QL): d [b, s], e [b, s] (e[:, i] = T[i+1][i]; the last column is ignored).
    Returns (lam [b, s] ascending, z [b, s, s] with row i = eigenvector i, flag [b]: bit 0 = a block of more than
    INVIT_KEEP + 1 close eigenvalues, bit 1 = QL did not converge)."""
    lib = hostsim()
    P = ctypes.c_void_p
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    b, s = d.shape
    lam, z, flag = np.zeros((b, s)), np.zeros((b, s, s)), np.zeros(b, np.int32)
    rc = lib.sympa_hostsim_tridiag_invit(P(d.ctypes.data), P(e.ctypes.data), ctypes.c_int64(b), s, P(lam.ctypes.data),
                                         P(z.ctypes.data), P(flag.ctypes.data))
    assert rc == 0
    return lam, z, flag


def graded_pairs(b, n, grade, seed=5):
    """Upper-model pairs whose E = L1^-1 (Z2 - Z1) L2^-T has singular values graded over 10^-grade (eigenvalues of H = E^H E over
    10^-2 grade): Z2 = Z1 + (1 + 0.3 i) L1 Q diag(s) Q^T L1^T, s_k = 10^(-grade k / (n - 1)).  numpy [b, 2, n, n] each."""
    g = torch.Generator().manual_seed(seed)
    z1 = torch.zeros(b, 2, n, n, dtype=torch.float64)
    z2 = torch.zeros_like(z1)
    for i in range(b):
        a = torch.randn(n, n, generator=g, dtype=torch.float64) * 0.3
        y1 = torch.eye(n, dtype=torch.float64) + a @ a.T
        x1 = torch.randn(n, n, generator=g, dtype=torch.float64)
        x1 = 0.5 * (x1 + x1.T)
        l1 = torch.linalg.cholesky(y1)
        q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
        s = torch.tensor([10.0 ** (-grade * k / (n - 1)) for k in range(n)], dtype=torch.float64)
        d = l1 @ (q * s) @ q.T @ l1.T
        z1[i, 0], z1[i, 1] = x1, y1
        z2[i, 0], z2[i, 1] = x1 + d, y1 + 0.3 * d
    return z1.numpy(), z2.numpy()
