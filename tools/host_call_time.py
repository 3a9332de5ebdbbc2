import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympa_amd import data, ops
from sympa_amd.model import Model
dev = torch.device("cuda:0")
class A:
    manifold, metric, dims, num_points = "upper", "riem", 4, 5041
    scale_coef, scale_init, train_scale = 1.0, 1.0, False
m = Model(A)
with torch.no_grad():
    m.embeddings.embeds.data = data.trained_like_table(5041, 4)
m = m.to(dev)
bl = [data.sample_pairs(5041, 65536, j).to(dev) for j in range(20)]
plan = m.prepare_batches(bl)
for _ in range(50): m.forward_batches(plan)
torch.cuda.synchronize()
for name, fn in (("plan", lambda: m.forward_batches(plan)), ("list", lambda: m.forward_batches(bl))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort()
    print(f"forward_batches({name}), K=20: host call median {ts[100][0]*1e6:.1f} us; call+sync median {sorted(t[1] for t in ts)[100]*1e6:.1f} us")
