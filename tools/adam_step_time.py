#!/usr/bin/env python3
"""RiemannianAdam training step (--optim radam, train.py:69-70): hipGraph replay against the eager step, headline shape.
   python tools/adam_step_time.py [model] [dims] [nodes] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sympa_amd import data
from sympa_amd.model import Model
from sympa_amd.optim import RiemannianAdam
from sympa_amd.train_step import GraphedTrainStep

model = sys.argv[1] if len(sys.argv) > 1 else "upper"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 5041
b = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dev = torch.device("cuda:0")


class A:
    manifold, metric, dims, num_points = model, "riem", n, nodes
    scale_coef, scale_init, train_scale = 1.0, 1.0, True


def make():
    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.trained_like_table(nodes, n, model=model, seed=1)
    return m.to(dev)


ids = data.sample_pairs(nodes, b, 0, 1).to(dev)
gd = (torch.rand(b, dtype=torch.float64) * 5 + 1).to(dev)
m1 = make()
o1 = RiemannianAdam(m1.parameters(), lr=1e-3, eps=1e-7, stabilize=None)
st = GraphedTrainStep(m1, o1, b, 50.0, dev)
mode = st.mode
for _ in range(3):
    st(ids, gd)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    st(ids, gd)
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / K
m2 = make()
o2 = RiemannianAdam(m2.parameters(), lr=1e-3, eps=1e-7, stabilize=None)
idc = ids.contiguous()


def eager():
    o2.zero_grad(set_to_none=False)
    m2.fused_loss_backward(idc, gd)
    torch.nn.utils.clip_grad_norm_(m2.parameters(), 50.0)
    o2.step()


for _ in range(3):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    eager()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / K
print(f"radam {model} n={n} nodes={nodes} batch={b}: graph replay [{mode}] {tg * 1e6:8.1f} us per training step | eager {te * 1e6:8.1f} us  ({te / tg:.2f}x)")
