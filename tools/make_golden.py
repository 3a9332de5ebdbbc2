#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED REFERENCE (fedelopez77/sympa at
/root/reference) in this container through tools/ref_shim.py.

Run here only (the GPU box has no /root/reference):   python tools/make_golden.py
Every fixture stores inputs AND outputs explicitly (never seeds), fp64.

Fixtures
  dist_{model}_n{n}.npz   z1,z2 [B,2,n,n] + reference dist for the 5 metrics + v (vector-valued
                          distance) for cases: init(1e-3) / s=0.1 / 0.5 / 1.0 / far(2.0, clamp
                          regime) / X=0 / diagonal / z1==z2 / tiny perturbation
  primitives_n{n}.npz     inverse, matrix_sqrt, cayley, inverse_cayley, takagi values (+ the
                          reconstruction property input), positive_conjugate_projection,
                          egrad2rgrad (upper, bounded), upper projx, intended bounded projx
  model_forward.npz       table + triplets + scale -> Model.forward output, composed from the
                          imported dist with an explicit gather (sympa.model itself needs real geoopt)
  autograd_{model}_n{n}.npz   d(sum(dist * coeff))/dz1,dz2 by torch autograd through the reference
  known_answers.json      numeric known-answer vectors held by the reference's own tests
                          (tests/test_math.py:175-305,411-427) stored as data
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
METRICS = ["riem", "fone", "finf", "fmin", "wsum"]


def sym(x):
    return 0.5 * (x + x.transpose(-1, -2))


def upper_points(b, n, s, g):
    """'trained-like' upper points: X = sym(N*s), Y = expm(sym(N*s))  (SURVEY 8d)."""
    x = sym(torch.randn(b, n, n, generator=g) * s)
    y = torch.matrix_exp(sym(torch.randn(b, n, n, generator=g) * s))
    return torch.stack((x, sym(y)), 1)


def init_points(b, n, g, eps=1e-3):
    """reference init distribution (upper_half.py:116-131)."""
    x = sym(torch.empty(b, n, n).uniform_(-eps, eps, generator=g))
    y = torch.eye(n).unsqueeze(0) + sym(torch.empty(b, n, n).uniform_(-eps, eps, generator=g))
    return torch.stack((x, y), 1)


def build_cases(n, g, b=24):
    cases = {}
    cases["init"] = (init_points(b, n, g), init_points(b, n, g))
    for s in (0.1, 0.5, 1.0):
        cases[f"s{s}"] = (upper_points(b, n, s, g), upper_points(b, n, s, g))
    if n <= 4:
        cases["far"] = (upper_points(b, n, 2.0, g), upper_points(b, n, 2.0, g))
    a, c = upper_points(b, n, 0.5, g), upper_points(b, n, 0.5, g)
    a[:, 0] = 0
    c[:, 0] = 0
    cases["xzero"] = (a, c)   # tests/test_upper_half.py:176-186
    a, c = upper_points(b, n, 0.5, g), upper_points(b, n, 0.5, g)
    eye = torch.eye(n).bool()
    a = torch.where(eye, a, torch.zeros_like(a))
    c = torch.where(eye, c, torch.zeros_like(c))
    cases["diag"] = (a, c)    # tests/test_upper_half.py:163-174
    a = upper_points(b, n, 0.5, g)
    cases["same"] = (a, a.clone())  # tests/test_upper_half.py:128-133
    a = init_points(b, n, g)
    c = a.clone()
    c[:, 0] = c[:, 0] * 1.001
    cases["perturb"] = (a, c)  # tests/test_upper_half.py:135-143
    return cases


def mp_exact_vvd(model, z1, z2, eps="1e-5", dps=50):
    """50-digit evaluation of the REFERENCE formula (sqrt, inverse, Cayley, singular values, clamp)
    with mpmath: tells fp64 rounding of the reference apart from real disagreement in the
    ill-conditioned 'far' regime (1 - d ~ 1e-5 .. 1e-8)."""
    import mpmath as mp
    mp.mp.dps = dps
    eps = mp.mpf(eps)
    out = []
    for a, b in zip(z1.numpy(), z2.numpy()):
        n = a.shape[-1]
        eye = mp.eye(n)
        A = mp.matrix(a[0].tolist()) + 1j * mp.matrix(a[1].tolist())
        B = mp.matrix(b[0].tolist()) + 1j * mp.matrix(b[1].tolist())
        if model == "bounded":   # cayley_transform.py:27-40
            A = 1j * (eye + A) * ((eye - A) ** -1)
            B = 1j * (eye + B) * ((eye - B) ** -1)
        X1 = A.apply(mp.re)
        Y1 = A.apply(mp.im)
        Y1 = (Y1 + Y1.T) / 2
        lam, V = mp.eigsy(Y1)
        Si = (V * mp.diag([mp.sqrt(l) for l in lam]) * V.T) ** -1
        Z3 = Si * (B - X1) * Si
        W = (Z3 - 1j * eye) * ((Z3 + 1j * eye) ** -1)
        sv = mp.svd_c(W, compute_uv=False)
        v = sorted(mp.log((1 + d) / max(1 - d, eps)) for d in sv)
        out.append([float(x) for x in v])
    return np.array(out)


def primitives_blob(n, g, sm, cay, tak, UH, BD):
    """inputs and reference outputs of the primitives / manifold methods at dimension n (drawn from g in a fixed order)."""
    b = 16
    zs = upper_points(b, n, 0.5, g)
    anyc = sm.to_symmetric(torch.randn(b, 2, n, n, generator=g))
    nonsym = torch.randn(b, 2, n, n, generator=g)
    realonly = anyc.clone(); realonly[:, 1] = 0
    imagonly = anyc.clone(); imagonly[:, 0] = 0
    prim = {
        "upper_pts": zs.numpy(),
        "csym": anyc.numpy(),
        "nonsym": nonsym.numpy(),
        "inverse_csym": sm.inverse(anyc).numpy(),
        "inverse_nonsym": sm.inverse(nonsym).numpy(),
        "inverse_realonly": sm.inverse(realonly).numpy(),
        "inverse_imagonly": sm.inverse(imagonly).numpy(),
        "matrix_sqrt_imag": sm.matrix_sqrt(sm.imag(zs)).numpy(),
        "cayley_upper": cay.cayley_transform(zs).numpy(),
        "bmm": sm.bmm(anyc, nonsym).numpy(),
        "bmm3": sm.bmm3(anyc, nonsym, anyc).numpy(),
        "compound": sm.to_compound_symmetric(anyc).numpy(),
        "takagi_values": tak.TakagiFactorization(n, return_eigenvectors=False).factorize(anyc).numpy(),
    }
    bounded_pts = cay.cayley_transform(zs)
    prim["inverse_cayley_of_cayley"] = cay.inverse_cayley_transform(bounded_pts).numpy()
    vals, s = tak.TakagiFactorization(n, return_eigenvectors=True).factorize(anyc)
    diag = sm.diag_embed(vals)
    prim["takagi_reconstruction"] = sm.bmm3(sm.conjugate(s), diag, sm.conj_trans(s)).numpy()
    # projection of symmetric matrices with some negative eigenvalues
    ysym = sym(torch.randn(b, n, n, generator=g))
    proj, keep = sm.positive_conjugate_projection(ysym)
    prim["pcp_in"] = ysym.numpy(); prim["pcp_out"] = proj.numpy(); prim["pcp_keep"] = keep.numpy()
    # manifold ops the optimiser calls
    u = torch.randn(b, 2, n, n, generator=g)
    up = UH(dims=n)
    bd = BD(dims=n)
    prim["grad_in"] = u.numpy()
    prim["upper_egrad2rgrad"] = up.egrad2rgrad(zs, u).numpy()
    prim["bounded_egrad2rgrad"] = bd.egrad2rgrad(bounded_pts, u).numpy()
    off = torch.randn(b, 2, n, n, generator=g)     # generic non-symmetric, Y indefinite
    prim["projx_in"] = off.numpy()
    prim["upper_projx"] = up.projx(off).numpy()
    # intended bounded projx (bounded_domain.py:55-84) via the eigenvector Takagi variant
    big = sm.to_symmetric(torch.randn(b, 2, n, n, generator=g))
    bvals, bs = tak.TakagiFactorization(n, return_eigenvectors=True).factorize(big)
    eps = 1e-5
    dtil = sm.diag_embed(torch.clamp(bvals, max=1 - eps))
    ztil = sm.bmm3(sm.conjugate(bs), dtil, sm.conj_trans(bs))
    keepb = torch.all(bvals < 1 - eps, dim=-1, keepdim=True)
    prim["bounded_projx_in"] = big.numpy()
    prim["bounded_projx"] = torch.where(keepb.unsqueeze(-1).unsqueeze(-1).expand_as(big), big, ztil).numpy()
    return prim


def main():
    torch.set_default_dtype(torch.float64)
    sm, cay, tak, UH, BD, met = ref_shim.import_reference()
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(20261002)

    for n in (2, 3, 4, 8):
        wsum_w = torch.linspace(-0.5, 1.5, n).reshape(1, n)  # includes a negative weight (relu)
        cases = build_cases(n, g)
        for model in ("upper", "bounded"):
            blob = {"wsum_weights": wsum_w.numpy(), "case_names": np.array(sorted(cases))}
            for name in sorted(cases):
                z1, z2 = cases[name]
                if model == "bounded":
                    z1, z2 = cay.cayley_transform(z1), cay.cayley_transform(z2)
                    # points must be exactly symmetric like a projected table row
                    z1, z2 = sm.to_symmetric(z1), sm.to_symmetric(z2)
                blob[f"{name}__z1"] = z1.numpy()
                blob[f"{name}__z2"] = z2.numpy()
                if name in ("far", "s1.0") and n <= 4:
                    blob[f"{name}__vvd_exact50"] = mp_exact_vvd(model, z1, z2)
                for metric in METRICS:
                    man = (UH if model == "upper" else BD)(dims=n, metric=met.MetricType.from_str(metric))
                    if metric == "wsum":
                        with torch.no_grad():
                            man.metric.weights.copy_(wsum_w)
                    with torch.no_grad():
                        d = man.dist(z1, z2)
                    blob[f"{name}__{metric}"] = d.detach().numpy()
            np.savez_compressed(os.path.join(OUT, f"dist_{model}_n{n}.npz"), **blob)

        # ---- primitives (pin the oracle function by function)
        np.savez_compressed(os.path.join(OUT, f"primitives_n{n}.npz"), **primitives_blob(n, g, sm, cay, tak, UH, BD))

        # ---- autograd goldens (for the backward kernel), moderate scale, distinct eigenvalues
        if n <= 4:
            for model in ("upper", "bounded"):
                blob = {}
                for metric in METRICS:
                    z1 = upper_points(12, n, 0.4, g)
                    z2 = upper_points(12, n, 0.4, g)
                    if model == "bounded":
                        z1 = sm.to_symmetric(cay.cayley_transform(z1))
                        z2 = sm.to_symmetric(cay.cayley_transform(z2))
                    z1.requires_grad_(True); z2.requires_grad_(True)
                    man = (UH if model == "upper" else BD)(dims=n, metric=met.MetricType.from_str(metric))
                    coeff = torch.rand(12, generator=g) + 0.5
                    out = man.dist(z1, z2)
                    (out * coeff).sum().backward()
                    blob[f"{metric}__z1"] = z1.detach().numpy(); blob[f"{metric}__z2"] = z2.detach().numpy()
                    blob[f"{metric}__coeff"] = coeff.numpy()
                    blob[f"{metric}__out"] = out.detach().numpy()
                    blob[f"{metric}__g1"] = z1.grad.numpy(); blob[f"{metric}__g2"] = z2.grad.numpy()
                    if metric == "wsum":
                        blob["wsum__gw"] = man.metric.weights.grad.numpy()
                np.savez_compressed(os.path.join(OUT, f"autograd_{model}_n{n}.npz"), **blob)

    # ---- Model.forward (model.py:16-41) composed from the imported dist + explicit gather
    n, N, b = 4, 50, 64
    table_u = upper_points(N, n, 0.3, g)
    table_b = sm.to_symmetric(cay.cayley_transform(upper_points(N, n, 0.3, g)))
    trip = torch.stack((torch.randint(0, N, (b,), generator=g), torch.randint(0, N, (b,), generator=g),
                        torch.randint(1, 9, (b,), generator=g)), 1)
    blob = {"table_upper": table_u.numpy(), "table_bounded": table_b.numpy(), "triplets": trip.numpy()}
    for scale, coef in ((1.0, 1.0), (0.05, 1.0), (3.0, 2.0)):   # 0.05 exercises clamp_min(0.1)
        sc = (torch.tensor([scale]) / coef).clamp_min(0.1)
        for model, table, cls in (("upper", table_u, UH), ("bounded", table_b, BD)):
            man = cls(dims=n, metric=met.MetricType.from_str("riem"))
            with torch.no_grad():
                out = man.dist(table[trip[:, 0]], table[trip[:, 1]]) * sc
            blob[f"{model}__scale{scale}_coef{coef}"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "model_forward.npz"), **blob)

    # ---- known-answer vectors held by the reference's own tests (data only)
    ka = {
        "_source": "numeric constants of /root/reference/tests/test_math.py (lines cited per entry)",
        "bmm": {"lines": "175-196",
                "x": [[[1, -3], [5, -7]], [[9, -11], [-14, 15]]],
                "y": [[[9, -11], [-14, 15]], [[1, -3], [5, -7]]],
                "expected": [[[97, -106], [82, -97]], [[221, -246], [-366, 413]]]},
        "bmm3": {"lines": "198-225",
                 "x": [[[1, -3], [5, -7]], [[9, -11], [-14, 15]]],
                 "y": [[[9, -11], [-14, 15]], [[1, -3], [5, -7]]],
                 "z": [[[-3, -1], [-2, 5]], [[-1, 3], [0, -2]]],
                 "expected": [[[142, -1782], [-418, 1357]], [[-268, -948], [190, 2871]]]},
        "inverse_symmetric_2d": {"lines": "227-242",
                                 "x": [[[1, -3], [-3, 7]], [[-9, 11], [11, 15]]],
                                 "expected": [[[256 / 8105, 141 / 16210], [141 / 16210, 23 / 16210]],
                                              [[921 / 16210, -356 / 8105], [-356 / 8105, -288 / 8105]]]},
        "inverse_symmetric_3d": {"lines": "244-264",
                                 "x": [[[-1, -3, 9], [-3, 5, 7], [9, 7, 11]], [[9, 4, -6], [4, 7, 9], [-6, 9, -3]]],
                                 "expected": [[[-36251 / 845665, -27631 / 845665, 188 / 9949],
                                               [-27631 / 845665, 251611 / 3382660, -689 / 39796],
                                               [188 / 9949, -689 / 39796, 1299 / 39796]],
                                              [[-18757 / 845665, -35642 / 845665, 532 / 9949],
                                               [-35642 / 845665, -112703 / 3382660, -1103 / 39796],
                                               [532 / 9949, -1103 / 39796, 289 / 39796]]]},
        "inverse_nonsymmetric_3d": {"lines": "266-287",
                                    "x": [[[-1, -3, 9], [3, 5, 7], [2, 9, 11]], [[9, 4, -6], [-4, 7, 9], [-2, 7, -3]]],
                                    "expected": [[[951 / 16589, 3223 / 16589, -4496 / 16589],
                                                  [5029 / 66356, 2486 / 16589, 1387 / 33178],
                                                  [2137 / 66356, 1030 / 16589, -1828 / 16589]],
                                                 [[-3143 / 16589, -3622 / 16589, 1532 / 16589],
                                                  [5533 / 66356, 6925 / 33178, -17977 / 66356],
                                                  [-4167 / 66356, -5779 / 33178, 6565 / 66356]]]},
        "pcp_positive": {"lines": "289-295", "x": [[0.9408, 0.1332], [0.1332, 0.5936]],
                         "expected": [[0.9408, 0.1332], [0.1332, 0.5936]]},
        "pcp_negative": {"lines": "297-305", "rtol": 1e-4, "x": [[6, 5], [5, 3]],
                         "expected": [[6.2566, 4.6551], [4.6551, 3.4636]]},
        "matrix_sqrt_4d": {"lines": "411-427", "rtol": 1e-5, "atol": 1e-6,
                           "x": [[0.7047, 0.2545, 0.0, -0.1481], [0.2545, 0.2122, 0.1481, 0.0],
                                 [0.0, 0.1481, 0.7047, 0.2545], [-0.1481, 0.0, 0.2545, 0.2122]],
                           "expected": [[0.802225, 0.213705, 0.0, -0.12436], [0.213705, 0.388671, 0.12436, 0],
                                        [0.0, 0.12436, 0.802225, 0.213705], [-0.12436, 0, 0.213705, 0.388671]]},
    }
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}  {os.path.getsize(os.path.join(OUT, fn))} B")


def autograd_goldens_large(dims=(6, 8)):
    """autograd_{model}_n{6,8}.npz: the same blobs as the n <= 4 autograd goldens, from their own generator (added after
    the first set was committed; `python tools/make_golden.py --autograd-large` writes only these files)."""
    torch.set_default_dtype(torch.float64)
    sm, cay, tak, UH, BD, met = ref_shim.import_reference()
    os.makedirs(OUT, exist_ok=True)
    for n in dims:
        g = torch.Generator().manual_seed(20261002 + 1000 * n)
        for model in ("upper", "bounded"):
            blob = {}
            for metric in METRICS:
                z1 = upper_points(10, n, 0.3, g)
                z2 = upper_points(10, n, 0.3, g)
                if model == "bounded":
                    z1 = sm.to_symmetric(cay.cayley_transform(z1))
                    z2 = sm.to_symmetric(cay.cayley_transform(z2))
                z1.requires_grad_(True); z2.requires_grad_(True)
                man = (UH if model == "upper" else BD)(dims=n, metric=met.MetricType.from_str(metric))
                coeff = torch.rand(10, generator=g) + 0.5
                out = man.dist(z1, z2)
                (out * coeff).sum().backward()
                blob[f"{metric}__z1"] = z1.detach().numpy(); blob[f"{metric}__z2"] = z2.detach().numpy()
                blob[f"{metric}__coeff"] = coeff.numpy()
                blob[f"{metric}__out"] = out.detach().numpy()
                blob[f"{metric}__g1"] = z1.grad.numpy(); blob[f"{metric}__g2"] = z2.grad.numpy()
                if metric == "wsum":
                    blob["wsum__gw"] = man.metric.weights.grad.numpy()
            path = os.path.join(OUT, f"autograd_{model}_n{n}.npz")
            np.savez_compressed(path, **blob)
            print(f"  {os.path.basename(path)}  {os.path.getsize(path)} B")


def dist_goldens_more(dims=(5, 6, 7, 12, 16)):
    """Round 3: `dist` of the imported reference at the dims whose kernels differ in KIND from those of n = 2, 3, 4, 8 --
    5..7 (Householder + lockstep QL in registers, the ring gather) and 12, 16 (sixteen lanes per pair) -- so that every
    kernel family is pinned by reference outputs directly, not only through the oracle.  Own RNG stream: the existing
    fixtures stay byte-identical.  Smaller batches at 12 / 16 keep the files small."""
    torch.set_default_dtype(torch.float64)
    sm, cay, tak, UH, BD, met = ref_shim.import_reference()
    g = torch.Generator().manual_seed(20261003)
    for n in dims:
        wsum_w = torch.linspace(-0.5, 1.5, n).reshape(1, n)
        cases = build_cases(n, g, b=24 if n <= 8 else 8)
        for model in ("upper", "bounded"):
            blob = {"wsum_weights": wsum_w.numpy(), "case_names": np.array(sorted(cases))}
            for name in sorted(cases):
                z1, z2 = cases[name]
                if model == "bounded":
                    z1, z2 = cay.cayley_transform(z1), cay.cayley_transform(z2)
                    z1, z2 = sm.to_symmetric(z1), sm.to_symmetric(z2)
                blob[f"{name}__z1"] = z1.numpy()
                blob[f"{name}__z2"] = z2.numpy()
                for metric in METRICS:
                    man = (UH if model == "upper" else BD)(dims=n, metric=met.MetricType.from_str(metric))
                    if metric == "wsum":
                        with torch.no_grad():
                            man.metric.weights.copy_(wsum_w)
                    with torch.no_grad():
                        d = man.dist(z1, z2)
                    blob[f"{name}__{metric}"] = d.detach().numpy()
            path = os.path.join(OUT, f"dist_{model}_n{n}.npz")
            np.savez_compressed(path, **blob)
            print(f"  {os.path.basename(path)}  {os.path.getsize(path)} B")


def goldens_round6():
    """Round 6 (review item 8): the holes left by the earlier sets, each from its own RNG stream so that every existing fixture
    stays byte-identical: reference-autograd goldens at n = 5 (QL with eigenvectors, one pair per lane) and n = 16 (sixteen lanes
    per pair); primitives at n = 5, 6, 7; the `far` (clamp regime, 1 - d ~ 1e-5 .. 1e-8) case of `dist` at n = 5..8, with the
    50-digit evaluation of the reference formula beside it (dist_far_{model}_n{n}.npz)."""
    torch.set_default_dtype(torch.float64)
    sm, cay, tak, UH, BD, met = ref_shim.import_reference()
    autograd_goldens_large(dims=(5, 16))
    for n in (5, 6, 7):
        g = torch.Generator().manual_seed(20261004 + 1000 * n)
        path = os.path.join(OUT, f"primitives_n{n}.npz")
        np.savez_compressed(path, **primitives_blob(n, g, sm, cay, tak, UH, BD))
        print(f"  {os.path.basename(path)}  {os.path.getsize(path)} B")
    for n in (5, 6, 7, 8):
        g = torch.Generator().manual_seed(20261005 + 1000 * n)
        wsum_w = torch.linspace(-0.5, 1.5, n).reshape(1, n)
        cases = {"far": (upper_points(12, n, 2.0, g), upper_points(12, n, 2.0, g)),
                 "far1.5": (upper_points(12, n, 1.5, g), upper_points(12, n, 1.5, g))}
        for model in ("upper", "bounded"):
            blob = {"wsum_weights": wsum_w.numpy(), "case_names": np.array(sorted(cases))}
            for name in sorted(cases):
                z1, z2 = cases[name]
                if model == "bounded":
                    z1, z2 = cay.cayley_transform(z1), cay.cayley_transform(z2)
                    z1, z2 = sm.to_symmetric(z1), sm.to_symmetric(z2)
                blob[f"{name}__z1"] = z1.numpy()
                blob[f"{name}__z2"] = z2.numpy()
                blob[f"{name}__vvd_exact50"] = mp_exact_vvd(model, z1, z2)
                for metric in METRICS:
                    man = (UH if model == "upper" else BD)(dims=n, metric=met.MetricType.from_str(metric))
                    if metric == "wsum":
                        with torch.no_grad():
                            man.metric.weights.copy_(wsum_w)
                    with torch.no_grad():
                        d = man.dist(z1, z2)
                    blob[f"{name}__{metric}"] = d.detach().numpy()
            path = os.path.join(OUT, f"dist_far_{model}_n{n}.npz")
            np.savez_compressed(path, **blob)
            print(f"  {os.path.basename(path)}  {os.path.getsize(path)} B")


if __name__ == "__main__":
    if "--round6" in sys.argv:
        goldens_round6()
    elif "--dist-more" in sys.argv:
        dist_goldens_more()
    elif "--autograd-more" in sys.argv:          # round 3: dims 7 (eight lanes per pair) and 12 (sixteen lanes per pair)
        autograd_goldens_large(dims=(7, 12))
    elif "--autograd-large" in sys.argv:
        autograd_goldens_large()
    else:
        main()
        autograd_goldens_large()
