for v in "" build_ab/n8_noql.so build_ab/n8_front.so; do
echo "== lib: ${v:-product}"
SYMPA_HIP_LIB=${v:+$PWD/$v} SYMPA_SELFCHECK=0 python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from sympa_amd import data, ops
dev = torch.device("cuda:0")
for n, nodes, b in ((8, 45500, 262144),):
    table = data.trained_like_table(nodes, n, seed=1).to(dev)
    pairs = data.sample_pairs(nodes, b, 0, 1).to(dev)
    gd = torch.rand(b, dtype=torch.float64, device=dev) * 5 + 1
    scale = torch.ones(1, dtype=torch.float64, device=dev)
    gt = torch.zeros_like(table); loss = torch.zeros(1, dtype=torch.float64, device=dev); gs = torch.zeros(1, dtype=torch.float64, device=dev)
    def step():
        return ops.model_loss_backward(table, pairs, gd, gt, loss, "upper", "riem", None, None, scale, gs, 1.0, 1.0)
    step(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    print(f"loss+backward (no zeroing) n={n} b={b}: {e0.elapsed_time(e1)*100:.1f} us")
    rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
    def step2():
        return ops.model_loss_backward_rows(table, pairs, gd, rows, loss, "upper", "riem", None, None, scale, gs, 1.0, 1.0)
    step2(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): step2()
    e1.record(); torch.cuda.synchronize()
    print(f"loss+backward per-pair rows   n={n} b={b}: {e0.elapsed_time(e1)*100:.1f} us")
PY
done
