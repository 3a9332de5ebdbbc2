import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import siegel_oracle as so
from tests.helpers import spd_points, rel_err
from sympa_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1616)
for b, s in ((1000, 0.3), (1000, 0.6), (1000, 1.0)):
    x, y = spd_points(b, 16, s, g), spd_points(b, 16, s, g)
    coop = ops.spd_dist_forward(x.to(dev), y.to(dev)).cpu()
    gen = ops.spd_dist_forward(x.to(dev), y.to(dev), flags=ops.FLAG_GENERIC).cpu()
    orc = so.spd_dist(x, y)
    lam = torch.linalg.eigvals(torch.linalg.solve(x, y)).real
    alt = torch.sqrt((torch.log(lam) ** 2).sum(-1))
    # Cholesky-based reference in fp64 torch: eig(L^-1 Y L^-T)
    L = torch.linalg.cholesky(x)
    A = torch.linalg.solve_triangular(L, y, upper=False)
    A = torch.linalg.solve_triangular(L, A.transpose(-1, -2), upper=False)
    ev = torch.linalg.eigvalsh(0.5 * (A + A.transpose(-1, -2)))
    ch = torch.sqrt((torch.log(ev) ** 2).sum(-1))
    r = lambda a, c: float(((a - c).abs() / c.abs()).max())
    print(s, "cond", float(torch.linalg.cond(x).max()), "coop-orc", r(coop, orc), "gen-orc", r(gen, orc), "coop-gen", r(coop, gen),
          "orc-alt", r(orc, alt), "orc-chol", r(orc, ch), "coop-chol", r(coop, ch), "gen-chol", r(gen, ch))
