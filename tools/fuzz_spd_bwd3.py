#!/usr/bin/env python3
"""Soak of the three-kernel SPD backward (n = 16; eigenvectors one pair per lane by inverse iteration) against the kernel that
accumulates QL rotations: random tables of many scales (near-identity ... ill-conditioned), random batch sizes, pairs with
exactly repeated or nearly repeated generalized eigenvalues mixed in; every gradient row compared.
    python tools/fuzz_spd_bwd3.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(99)
n = 16
os.environ["SYMPA_SPD_BWD_WORKSPACE_MIN"] = "1"


def sym(a):
    return 0.5 * (a + a.transpose(-1, -2))


def points(b, s):
    return sym(torch.matrix_exp(sym(torch.randn(b, n, n, generator=g, dtype=torch.float64) * s)))


t0 = time.time()
cases = pairs = 0
worst_row = worst_dist = 0.0
while time.time() - t0 < budget:
    b = int(torch.randint(1, 30000, (1,), generator=g))
    s = float(10.0 ** (torch.rand(1, generator=g) * 3.3 - 3.3))          # 5e-4 .. 1
    x, y = points(b, s), points(b, s)
    k = max(1, b // 50)
    idx = torch.randint(0, b, (k,), generator=g)
    mode = cases % 4
    if mode == 1:            # y = L (I + Q diag(clustered) Q^T) L^T: blocks of 2..3 (served) and 5..7 (handed back) close eigenvalues
        lam = torch.sort(torch.randn(k, n, generator=g, dtype=torch.float64) * s, dim=1).values
        size = int(torch.randint(2, 8, (1,), generator=g))
        gap = float(10.0 ** (-torch.rand(1, generator=g) * 12 - 2))
        for j in range(1, size):
            lam[:, 3 + j] = lam[:, 3] + gap * j * lam.abs().max(1).values
        lam = lam.clamp_min(-0.9)
        q, _ = torch.linalg.qr(torch.randn(k, n, n, generator=g, dtype=torch.float64))
        l = torch.linalg.cholesky(x[idx])
        y[idx] = sym(l @ (torch.eye(n, dtype=torch.float64) + (q * lam[:, None, :]) @ q.transpose(-1, -2)) @ l.transpose(-1, -2))
    elif mode == 2:
        y[idx] = x[idx] * (1.0 + torch.rand(k, 1, 1, generator=g, dtype=torch.float64))       # all eigenvalues equal
    elif mode == 3:
        y[idx] = x[idx]                                                                       # zero distance
    go = (torch.rand(b, generator=g, dtype=torch.float64) + 0.5).to(dev)
    xd, yd = x.to(dev), y.to(dev)
    os.environ.pop("SYMPA_SPD_BWD_NO_WORKSPACE", None)
    rows_n, out_n = ops.spd_backward_rows(xd, yd, grad_out=go, want_out=True)
    os.environ["SYMPA_SPD_BWD_NO_WORKSPACE"] = "1"
    rows_o, out_o = ops.spd_backward_rows(xd, yd, grad_out=go, want_out=True)
    os.environ.pop("SYMPA_SPD_BWD_NO_WORKSPACE")
    ops.check_status(dev)
    assert torch.isfinite(rows_n).all() and torch.isfinite(out_n).all()
    scale_ = rows_o.abs().reshape(2 * b, -1).max(1).values.clamp_min(1e-300)
    err = float(((rows_n - rows_o).abs().reshape(2 * b, -1).max(1).values / scale_).max())
    derr = float(((out_n - out_o).abs() / out_o.abs().clamp_min(1e-12)).max())
    worst_row, worst_dist = max(worst_row, err), max(worst_dist, derr)
    if not (err < 1e-6 and derr < 1e-9):
        print(f"MISMATCH case {cases} mode {mode} b={b} s={s:.2e}: row err {err:.2e} dist err {derr:.2e}")
        sys.exit(1)
    cases += 1
    pairs += b
print(f"fuzz_spd_bwd3 ok: {cases} cases, {pairs} pairs in {time.time() - t0:.0f} s: worst row error {worst_row:.2e} (relative to the row's largest "
      f"entry), worst distance error {worst_dist:.2e}")
