"""CPU oracle for the Siegel-distance hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a torch-CPU fp64 *restatement* of the reference's algorithm for
`Model.forward -> manifold.dist` (fedelopez77/sympa).  It follows the reference op for op
(same sequence of eigh / inverse / matmul / cat / log / norm calls) so that it can serve as

  * the checker the HIP path is compared against in `tests/`, `__graft_entry__.smoke()`,
  * the timed `cpu_baseline` leg of `bench.py` (kind = "port").

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it.
The product package (`sympa_amd/`) never imports anything from `oracle/`.

Parity pin: PINNED.  `tests/golden/*.npz` hold inputs+outputs produced by the *imported
reference itself* (tools/make_golden.py, through tools/ref_shim.py) plus the known-answer
vectors of the reference's own tests (tests/test_math.py numeric constants, stored as data);
`tests/test_oracle_golden.py` checks every function below against them (<= 1e-12 rel).

Third-party arithmetic not under /root/reference that this restates through torch:
  * torch.symeig (torch==1.5.1 pinned by the reference README.md:38, default upper=True)
      -> torch.linalg.eigh(UPLO="U")  (same LAPACK syevd family; eigenvalues ascending)
  * geoopt.linalg.batch_linalg.sym (geoopt>=0.3.1, README.md:40) -> 0.5*(x + x^T)

Layout convention (reference csym_math.py:1-8): a batch of complex n x n matrices is a real
tensor [b, 2, n, n]; index 0 = real part, index 1 = imaginary part.
"""
from __future__ import annotations

import torch

# reference sympa/config.py:17-21
EPS = {torch.float32: 4e-3, torch.float64: 1e-5}
INIT_EPS = 1e-3

METRICS = ("riem", "fone", "finf", "fmin", "wsum")  # reference manifolds/metrics.py:6-12


# --------------------------------------------------------------------------- csym_math
def re(z):  # csym_math.py:15-21
    return z[:, 0]


def im(z):  # csym_math.py:24-30
    return z[:, 1]


def pack(a, b):  # "stick", csym_math.py:33-41
    return torch.stack((a, b), dim=1)


def sym(x):  # geoopt.linalg.batch_linalg.sym, used at csym_math.py:12,136-137
    return 0.5 * (x + x.transpose(-1, -2))


def to_symmetric(z):  # csym_math.py:131-138
    return pack(sym(re(z)), sym(im(z)))


def conjugate(z):  # csym_math.py:44-46
    return pack(re(z), -im(z))


def conj_trans(z):  # csym_math.py:53-55
    return conjugate(z).transpose(-1, -2)


def cmatmul(x, y):
    """Complex product as four real batched matmuls (csym_math.py:91-115)."""
    a, b = re(x), im(x)
    c, d = re(y), im(y)
    ac = a @ c
    bd = b @ d
    ad = a @ d
    bc = b @ c
    return pack(ac - bd, ad + bc)


def cmatmul3(x, y, z):  # csym_math.py:118-128
    return cmatmul(cmatmul(x, y), z)


def identity_like(z):  # csym_math.py:334-343
    b, _, _, n = z.shape
    eye = torch.eye(n, dtype=z.dtype, device=z.device)
    out = torch.zeros(b, 2, n, n, dtype=z.dtype, device=z.device)
    out[:, 0] = eye
    return out


def times_i(z):  # csym_math.py:164-169
    return pack(-im(z), re(z))


def symeig(y):
    """Stand-in for torch.symeig(y, eigenvectors=True) (csym_math.py:281-292)."""
    return torch.linalg.eigh(y, UPLO="U")


def matrix_sqrt(y):  # csym_math.py:509-520
    lam, vec = symeig(y)
    return vec @ torch.diag_embed(torch.sqrt(lam)) @ vec.transpose(-1, -2)


def cinverse(z):
    """Complex inverse by two real inversions (Falkenberg), incl. the reference's
    first-row-all-zero special cases (csym_math.py:197-249)."""
    a, c = re(z), im(z)
    n = a.size(-1)
    real_row0_zero = (a[:, 0] == 0).to(a.dtype).sum(-1) == n
    imag_row0_zero = (c[:, 0] == 0).to(c.dtype).sum(-1) == n
    any_real_zero = bool(torch.any(real_row0_zero))
    any_imag_zero = bool(torch.any(imag_row0_zero))

    if any_real_zero:
        inv_c_saved = torch.inverse(c)
        bump = torch.zeros_like(a)
        bump[real_row0_zero] = torch.eye(n, dtype=a.dtype, device=a.device)
        a = a + bump
    if any_imag_zero:
        inv_a_saved = torch.inverse(a)
        bump = torch.zeros_like(c)
        bump[imag_row0_zero] = torch.eye(n, dtype=c.dtype, device=c.device)
        c = c + bump

    r = torch.inverse(a) @ c
    u = torch.inverse(c @ r + a)
    v = (r @ u) * -1

    if any_real_zero:
        m = real_row0_zero.reshape(-1, 1, 1)
        u = torch.where(m, torch.zeros_like(u), u)
        v = torch.where(m, inv_c_saved, v)
    if any_imag_zero:
        m = imag_row0_zero.reshape(-1, 1, 1)
        u = torch.where(m, inv_a_saved, u)
        v = torch.where(m, torch.zeros_like(v), v)
    return pack(u, v)


def positive_conjugate_projection(y):  # csym_math.py:252-278
    lam, s = symeig(y)
    lam_c = torch.clamp(lam, min=EPS[y.dtype])
    y_tilde = s @ torch.diag_embed(lam_c) @ s.transpose(-1, -2)
    keep = torch.all(lam > EPS[y.dtype], dim=-1, keepdim=True)
    return torch.where(keep.unsqueeze(-1).expand_as(y), y, y_tilde), keep


def compound_symmetric(z):  # csym_math.py:421-436   M = [[A, B], [B, -A]]
    a, b = re(z), im(z)
    top = torch.cat((a, b), dim=-1)
    bot = torch.cat((b, -a), dim=-1)
    return torch.cat((top, bot), dim=-2)


# --------------------------------------------------------------------------- cayley
def cayley_transform(z):  # cayley_transform.py:10-24   (Z - iI)(Z + iI)^-1
    ident = identity_like(z)
    i_ident = pack(im(ident), re(ident))
    return cmatmul(z - i_ident, cinverse(z + i_ident))


def inverse_cayley_transform(z):  # cayley_transform.py:27-40   i(I + Z)(I - Z)^-1
    ident = identity_like(z)
    return cmatmul(times_i(ident + z), cinverse(ident - z))


# --------------------------------------------------------------------------- takagi
def takagi_values(z):
    """Singular values (ascending) of complex-symmetric z via the real 2n x 2n compound
    matrix (takagi_factorization.py:66-75)."""
    lam, _ = symeig(compound_symmetric(z))
    return lam[:, z.shape[-1]:]


def takagi_factorize(z):
    """Values and unitary S with z = conj(S) D S^H (takagi_factorization.py:45-64)."""
    lam, q = symeig(compound_symmetric(z))
    n = z.shape[-1]
    right = q[..., n:]
    u = pack(right[..., :n, :], -right[..., n:, :])
    return lam[:, n:], u


# --------------------------------------------------------------------------- metrics
def fmin_weights(n, dtype=torch.float64):  # metrics.py:86-89  -> [0, 2, ..., 2(n-1)]
    return (2 * (n + 1 - torch.arange(n + 1, 1, -1))).to(dtype).unsqueeze(0)


def compute_metric(v, metric: str, weights=None):
    """v: [b, n] ascending vector-valued distance -> [b] (metrics.py:42-121)."""
    if metric == "riem":
        return torch.norm(v, dim=-1)
    if metric == "fone":
        return torch.sum(v, dim=-1)
    if metric == "finf":
        return v[:, -1]
    if metric == "fmin":
        return torch.sum(fmin_weights(v.shape[-1], v.dtype) * v, dim=-1)
    if metric == "wsum":
        w = torch.ones(1, v.shape[-1], dtype=v.dtype) if weights is None else weights.reshape(1, -1)
        return torch.sum(torch.nn.functional.relu(w) * v, dim=-1)
    raise KeyError(metric)


# --------------------------------------------------------------------------- dist
def vector_valued_distance(z1, z2, check=True):
    """v_i = log((1+d_i)/clamp(1-d_i, eps)), d = Takagi values of the Cayley image of
    Y1^-1/2 (Z2 - X1) Y1^-1/2   (siegel_manifold.py:41-70). Returns (v [b,n], d [b,n])."""
    x1, y1 = re(z1), im(z1)
    inv_sqrt_y1 = matrix_sqrt(y1).inverse()
    inv_sqrt_y1 = pack(inv_sqrt_y1, torch.zeros_like(inv_sqrt_y1))
    shifted = z2 - pack(x1, torch.zeros_like(x1))
    z3 = cmatmul3(inv_sqrt_y1, shifted, inv_sqrt_y1)
    w = cayley_transform(z3)
    d = takagi_values(w)
    eps = EPS[d.dtype]
    if check:  # siegel_manifold.py:64-66
        assert torch.all(d >= 0 - eps), f"Eigenvalues: {d}"
        assert torch.all(d <= 1.01), f"Eigenvalues: {d}"
    v = torch.log((1 + d) / (1 - d).clamp(min=eps))
    return v, d


def upper_dist(z1, z2, metric="riem", weights=None, check=True):
    """UpperHalfManifold.dist == SiegelManifold.dist (siegel_manifold.py:41-72)."""
    v, _ = vector_valued_distance(z1, z2, check=check)
    return compute_metric(v, metric, weights)


def bounded_dist(z1, z2, metric="riem", weights=None, check=True):
    """BoundedDomainManifold.dist (bounded_domain.py:27-39)."""
    return upper_dist(inverse_cayley_transform(z1), inverse_cayley_transform(z2), metric, weights, check)


def manifold_dist(model: str, z1, z2, metric="riem", weights=None, check=True):
    if model == "upper":
        return upper_dist(z1, z2, metric, weights, check)
    if model == "bounded":
        return bounded_dist(z1, z2, metric, weights, check)
    raise KeyError(model)


# --------------------------------------------------------------------------- Model.forward
def get_scale(scale, scale_coef):  # model.py:40-41
    return (scale / scale_coef).clamp_min(0.1)


def model_forward(table, triplets, model="upper", metric="riem", weights=None,
                  scale=None, scale_coef=1.0, check=True):
    """Model.forward (model.py:16-30): gather two rows per triplet (embeddings.py:29-34),
    manifold.dist, times clamp_min(scale/scale_coef, 0.1)."""
    src, dst = triplets[:, 0], triplets[:, 1]
    d = manifold_dist(model, table[src], table[dst], metric, weights, check)
    if scale is None:
        scale = torch.tensor([scale_coef * 1.0], dtype=table.dtype)
    return d * get_scale(scale, scale_coef)


# --------------------------------------------------------------------------- manifold ops used by the optimiser
def upper_egrad2rgrad(z, u):  # upper_half.py:25-40
    y = im(z)
    return pack(y.bmm(re(u)).bmm(y), y.bmm(im(u)).bmm(y))


def upper_projx(z):  # siegel_manifold.py:130-137 + upper_half.py:42-66
    z = to_symmetric(z)
    y_tilde, keep = positive_conjugate_projection(im(z))
    return pack(re(z), y_tilde), keep


def id_minus_conj_z_z(z):  # bounded_domain.py:163-170
    return identity_like(z) - cmatmul(conjugate(z), z)


def bounded_egrad2rgrad(z, u):  # bounded_domain.py:41-53
    a = id_minus_conj_z_z(z)
    return cmatmul3(a, u, a)


def bounded_projx(z):
    """The *intended* BoundedDomainManifold.projx (bounded_domain.py:55-84).  At the surveyed
    commit the in-tree call is broken (the manifold builds its Takagi object with
    return_eigenvectors=False); this restates it with the eigenvector variant."""
    z = to_symmetric(z)
    lam, s = takagi_factorize(z)
    lam_c = torch.clamp(lam, max=1 - EPS[z.dtype])
    diag = torch.diag_embed(lam_c)
    diag = pack(diag, torch.zeros_like(diag))
    z_tilde = cmatmul3(conjugate(s), diag, conj_trans(s))
    keep = torch.all(lam < 1 - EPS[z.dtype], dim=-1, keepdim=True)
    mask = keep.unsqueeze(-1).unsqueeze(-1).expand_as(z)
    return torch.where(mask, z, z_tilde), keep


def rsgd_step(model, table, grad, lr, weight_decay=0.0):
    """geoopt.optim.RiemannianSGD.step, momentum 0 (the optimiser of train.py:66-68; geoopt >=0.3.1 is an
    un-vendored dependency, algorithm restated from geoopt/optim/rsgd.py):
        point <- retr(point, -lr * egrad2rgrad(point, grad + weight_decay * point)),  retr = projx(x + u)."""
    g = grad + weight_decay * table
    if model == "upper":
        return upper_projx(table - lr * upper_egrad2rgrad(table, g))
    return bounded_projx(table - lr * bounded_egrad2rgrad(table, g))


def upper_random(n_points, dims, from_=-INIT_EPS, to=INIT_EPS, generator=None):
    """Distribution of UpperHalfManifold.random (upper_half.py:116-131)."""
    pert = sym(torch.empty(n_points, dims, dims, dtype=torch.float64).uniform_(from_, to, generator=generator))
    y = torch.eye(dims, dtype=torch.float64).unsqueeze(0) + pert
    x = sym(torch.empty(n_points, dims, dims, dtype=torch.float64).uniform_(from_, to, generator=generator))
    return pack(x, y)


def distortion_loss(graph_d, manifold_d):  # losses.py:10-19
    return torch.abs(torch.pow(manifold_d / graph_d, 2) - 1).sum()


# --------------------------------------------------------------------------- spd (PARITY UNPINNED)
def _sym_funcm(x, func):
    """geoopt.linalg.batch_linalg.sym_funcm: apply `func` to the eigenvalues of a symmetric matrix."""
    lam, v = torch.linalg.eigh(x, UPLO="U")
    return v @ torch.diag_embed(func(lam)) @ v.transpose(-1, -2)


def spd_dist(x, y):
    """geoopt.manifolds.SymmetricPositiveDefinite.dist with the default AIM metric, restated from the published
    source (geoopt/manifolds/symmetric_positive_definite.py): || sym_logm(x^-1/2 y x^-1/2) ||_F with the inverse
    square root and the logarithm taken through eigh.  geoopt is NOT in the reference tree (un-vendored,
    >=0.3.1, README.md:40) and not installed: this function is pinned by nothing but its own formula."""
    inv_sqrt_x = _sym_funcm(x, lambda lam: torch.rsqrt(lam))
    inner = inv_sqrt_x @ y @ inv_sqrt_x
    return torch.norm(_sym_funcm(inner, torch.log), dim=[-1, -2])


def spd_model_forward(table, triplets, scale=None, scale_coef=1.0):
    d = spd_dist(table[triplets[:, 0]], table[triplets[:, 1]])
    if scale is None:
        scale = torch.tensor([scale_coef * 1.0], dtype=table.dtype)
    return d * get_scale(scale, scale_coef)


def spd_egrad2rgrad(x, u):
    """geoopt SymmetricPositiveDefinite.egrad2rgrad: x @ proju(x, u) @ x with proju = sym (restated; unpinned)."""
    return x @ sym(u) @ x


def spd_retr(x, u):
    """geoopt SymmetricPositiveDefinite.retr: sym(x + u + 1/2 u x^-1 u) (restated; unpinned)."""
    return sym(x + u + 0.5 * u @ torch.linalg.inv(x) @ u)


def spd_projx(x):
    """geoopt SymmetricPositiveDefinite.projx: sym_funcm(sym(x), abs) (restated; unpinned)."""
    return _sym_funcm(sym(x), torch.abs)


def spd_rsgd_step(table, grad, lr, weight_decay=0.0):
    """geoopt.optim.RiemannianSGD.step (momentum 0, stabilize None) on an spd table."""
    g = grad + weight_decay * table
    return spd_retr(table, -lr * spd_egrad2rgrad(table, g))


def upper_inner(z, u, v=None):
    """UpperHalfManifold.inner (upper_half.py:68-91): Re tr[y^-1 u y^-1 conj(v)] -> [b]."""
    v = u if v is None else v
    iy = torch.linalg.inv(im(z))
    iyc = pack(iy, torch.zeros_like(iy))
    res = cmatmul(cmatmul3(iyc, u, iyc), conjugate(v))
    return torch.diagonal(re(res), dim1=-2, dim2=-1).sum(-1)


def bounded_inner(z, u, v=None):
    """BoundedDomainManifold.inner (bounded_domain.py:86-116): Re tr[(I - conj(z) z)^-1 u (I - z conj(z))^-1 conj(v)]."""
    v = u if v is None else v
    ident = identity_like(z)
    a = cinverse(ident - cmatmul(conjugate(z), z))
    b = cinverse(ident - cmatmul(z, conjugate(z)))
    res = cmatmul(cmatmul3(a, u, b), conjugate(v))
    return torch.diagonal(re(res), dim1=-2, dim2=-1).sum(-1)


def radam_step(model, table, grad, state, lr, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.0):
    """geoopt.optim.RiemannianAdam.step for a Siegel table (restated from geoopt/optim/radam.py; geoopt is absent):
    state = {"step", "exp_avg" [N,2,n,n], "exp_avg_sq" [N]}; transp is the identity (siegel_manifold.py:142-154)."""
    b1, b2 = betas
    state["step"] += 1
    g = grad + weight_decay * table
    g = upper_egrad2rgrad(table, g) if model == "upper" else bounded_egrad2rgrad(table, g)
    state["exp_avg"] = b1 * state["exp_avg"] + (1 - b1) * g
    inn = upper_inner(table, g) if model == "upper" else bounded_inner(table, g)
    state["exp_avg_sq"] = b2 * state["exp_avg_sq"] + (1 - b2) * inn
    bc1, bc2 = 1 - b1 ** state["step"], 1 - b2 ** state["step"]
    denom = (state["exp_avg_sq"] / bc2).sqrt() + eps
    direction = (state["exp_avg"] / bc1) / denom.view(-1, 1, 1, 1)
    new = table - lr * direction
    return (upper_projx(new) if model == "upper" else bounded_projx(new))[0]
