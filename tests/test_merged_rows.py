"""Round 5: the deterministic rows form with runs of equal source ids MERGED inside a wave (C-ABI SYMPA_FLAG_MERGE_SRC,
csrc/siegel_bwd_kernel.hpp::store_rows_merged; ops.sorted_slots(..., merged_src=b)).  The training loop of the reference adds the
gradient rows of a batch in whatever order autograd's index_add takes (runner.py:98-118); the deterministic mode fixes that order, and
merging changes WHICH fixed order (the rows of a run first) -- so it equals the plain rows form to rounding, not bitwise, and is itself
bitwise reproducible."""
import pytest
import torch

from tests.helpers import points


def test_sorted_slots_leaves_out_the_unwritten_source_slots():
    from sympa_amd import ops
    b, rows = 200, 7
    g = torch.Generator().manual_seed(1)
    src = torch.sort(torch.randint(0, rows, (3, b), generator=g), dim=1).values
    src[1] = torch.randint(0, rows, (b,), generator=g)                       # an unsorted batch: runs of length one (mostly)
    dst = torch.randint(0, rows, (3, b), generator=g)
    keys = torch.cat((src, dst), 1)
    order, rowptr = ops.sorted_slots(keys, rows, merged_src=b)
    for s in range(3):
        ends = [k for k in range(b) if k == b - 1 or k % 64 == 63 or src[s, k + 1] != src[s, k]]
        want = {r: [k for k in ends if src[s, k] == r] + [b + k for k in range(b) if dst[s, k] == r] for r in range(rows)}
        for r in range(rows):
            got = order[s, rowptr[s, r]:rowptr[s, r + 1]].tolist()
            assert got == want[r], (s, r)
        assert int(rowptr[s, rows]) == len(ends) + b
    plain = ops.sorted_slots(keys, rows)
    assert int(plain[1][0, rows]) == 2 * b


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("model,n,b", [("upper", 4, 4096), ("upper", 4, 1000), ("bounded", 4, 700), ("upper", 5, 333), ("upper", 6, 130),
                                       ("bounded", 6, 64), ("upper", 2, 65), ("upper", 3, 1)])
@pytest.mark.parametrize("sort", [True, False])
def test_gpu_merged_rows_equal_the_plain_rows_form(dev, model, n, b, sort):
    from sympa_amd import data, ops
    nodes = 50
    g = torch.Generator().manual_seed(17 * n + b)
    table = points(model, nodes, n, 0.3, g).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (b,), generator=g), torch.randint(0, nodes, (b,), generator=g),
                        torch.randint(1, 9, (b,), generator=g)), 1)
    if sort:
        trip = data.sort_batches_by_source(trip, b)
    trip = trip.to(dev)
    gd = trip[:, 2].to(torch.float64).contiguous()
    scale = torch.full((1,), 1.2, dtype=torch.float64, device=dev)
    metric = "wsum" if n >= 4 else "riem"
    w = (torch.rand(n, generator=g, dtype=torch.float64) + 0.1).to(dev)
    res = {}
    for name, flags in (("plain", 0), ("merged", ops.FLAG_MERGE_SRC), ("merged2", ops.FLAG_MERGE_SRC)):
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        gs = torch.zeros(1, dtype=torch.float64, device=dev)
        gw = torch.zeros(n, dtype=torch.float64, device=dev)
        rows = torch.full((2 * b, 2, n, n), float("nan"), dtype=torch.float64, device=dev)     # unwritten slots stay NaN
        wp = torch.zeros((b + 63) // 64, 2 + n, dtype=torch.float64, device=dev)
        ops.model_train_backward(table, trip, gd, b, loss, model, metric, w, gw, scale, gs, 1.0, 1.0, grad_rows=rows,
                                 wave_partials=wp, flags=flags)
        order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), nodes, merged_src=b if flags else 0)
        grad = torch.zeros_like(table)
        ops.segment_sum_rows_(grad, rows, order, rowptr, wave_partials=wp, num_waves=(b + 63) // 64, partial_stride=2 + n,
                              loss=loss, grad_scale=gs, grad_weights=gw)
        res[name] = (grad.cpu(), loss.cpu(), gs.cpu(), gw.cpu(), int(rowptr[0, nodes]), rows.cpu())
    ops.check_status(dev)
    assert torch.isfinite(res["merged"][0]).all()
    scale_ = float(res["plain"][0].abs().max())
    assert float((res["merged"][0] - res["plain"][0]).abs().max()) <= 1e-13 * scale_
    for k in (1, 2, 3):
        assert torch.equal(res["merged"][k], res["plain"][k])                  # the per-wave sums do not depend on the rows form
    assert torch.equal(res["merged"][0], res["merged2"][0])                    # bitwise reproducible
    if sort and b >= 64:
        assert res["merged"][4] < res["plain"][4]                              # fewer slots in the lists
    # a slot is written iff it ends a run: the others still hold the NaN fill (source half), every target slot is written
    written = torch.isfinite(res["merged"][5][:b].reshape(b, -1)).all(1)
    src = trip[:, 0].cpu()
    ends = torch.ones(b, dtype=torch.bool)
    ends[:-1] = (src[1:] != src[:-1]) | ((torch.arange(b - 1) & 63) == 63)
    assert torch.equal(written, ends)
    assert torch.isfinite(res["merged"][5][b:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("model,metric", [("upper", "riem"), ("bounded", "finf")])
def test_gpu_graphed_deterministic_step_with_merged_rows(dev, model, metric, monkeypatch):
    """GraphedTrainStep(deterministic=True) at dims 4: batches sorted by source + merged rows (the default) against the same epoch
    with SYMPA_NO_MERGE_SRC=1 -- the tables after three steps agree to rounding, the merged run is bitwise reproducible."""
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianSGD
    from sympa_amd.train_step import GraphedTrainStep, merges_source_rows
    n, nodes, b, steps = 4, 300, 4096, 3
    g = torch.Generator().manual_seed(9)
    trip = torch.stack((torch.randint(0, nodes, (b * steps,), generator=g), torch.randint(0, nodes, (b * steps,), generator=g),
                        torch.randint(1, 9, (b * steps,), generator=g)), 1).to(dev)

    def run():
        class A:
            manifold, dims, num_points = model, n, nodes
            scale_coef, scale_init, train_scale = 1.0, 1.0, True
        A.metric = metric
        torch.manual_seed(3)
        m = Model(A)
        with torch.no_grad():
            m.embeddings.embeds.data = points(model, nodes, n, 0.3, torch.Generator().manual_seed(4))
        m = m.to(dev)
        opt = RiemannianSGD(m.parameters(), lr=1e-2, stabilize=None)
        step = GraphedTrainStep(m, opt, b, 50.0, dev, deterministic=True)
        assert step.load_epoch(trip) == steps
        step.run_steps(steps)
        torch.cuda.synchronize()
        return m.embeddings.embeds.detach().cpu().clone(), merges_source_rows(m, True, b)

    merged, on = run()
    assert on
    again, _ = run()
    assert torch.equal(merged, again)
    monkeypatch.setenv("SYMPA_NO_MERGE_SRC", "1")
    plain, off = run()
    assert not off
    assert float((merged - plain).abs().max()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("model,n,b", [("upper", 4, 4096), ("bounded", 4, 1000), ("upper", 5, 333), ("bounded", 6, 130), ("upper", 2, 65),
                                       ("upper", 3, 1)])
@pytest.mark.parametrize("sort", [True, False])
def test_gpu_merged_atomic_scatter_equals_the_plain_scatter(dev, model, n, b, sort):
    """The scatter form with SYMPA_FLAG_MERGE_SRC (runs summed in the wave's LDS tile, one atomic row per run) against the plain
    scatter: the same table gradient to rounding (fp64 atomics are unordered either way), loss and scalar gradients too."""
    from sympa_amd import data, ops
    nodes = 40
    g = torch.Generator().manual_seed(23 * n + b)
    table = points(model, nodes, n, 0.3, g).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (b,), generator=g), torch.randint(0, nodes, (b,), generator=g),
                        torch.randint(1, 9, (b,), generator=g)), 1)
    if sort:
        trip = data.sort_batches_by_source(trip, b)
    trip[min(b - 1, 3), 0] = trip[0, 0]                       # (a run of its own in the middle of another row's run)
    trip = trip.to(dev)
    gd = trip[:, 2].to(torch.float64).contiguous()
    scale = torch.full((1,), 0.9, dtype=torch.float64, device=dev)
    res = {}
    for name, flags in (("plain", 0), ("merged", ops.FLAG_MERGE_SRC)):
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        gs = torch.zeros(1, dtype=torch.float64, device=dev)
        grad = torch.zeros_like(table)
        ops.model_train_backward(table, trip, gd, b, loss, model, "riem", None, None, scale, gs, 1.0, 1.0, grad_table=grad, flags=flags)
        res[name] = (grad.cpu(), loss.cpu(), gs.cpu())
    ops.check_status(dev)
    big = float(res["plain"][0].abs().max())
    assert float((res["merged"][0] - res["plain"][0]).abs().max()) <= 1e-12 * big
    assert abs(float(res["merged"][1] - res["plain"][1])) <= 1e-12 * abs(float(res["plain"][1]))
    assert abs(float(res["merged"][2] - res["plain"][2])) <= 1e-11 * abs(float(res["plain"][2]))


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["rows", "scatter"])
def test_gpu_merged_forms_with_an_out_of_range_id(dev, form):
    """An out-of-range source id inside a run: the pair contributes nothing and the status word raises IndexError, in both merged
    forms as in the plain ones; the rest of the gradient is unchanged."""
    from sympa_amd import data, ops
    model, n, b, nodes = "upper", 4, 300, 20
    g = torch.Generator().manual_seed(5)
    table = points(model, nodes, n, 0.3, g).to(dev)
    trip = torch.stack((torch.randint(0, nodes, (b,), generator=g), torch.randint(0, nodes, (b,), generator=g),
                        torch.randint(1, 9, (b,), generator=g)), 1)
    trip = data.sort_batches_by_source(trip, b)
    trip[70, 0] = nodes + 3                                   # in the middle of a run of the second wave
    trip[200, 1] = -1
    trip = trip.to(dev)
    gd = trip[:, 2].to(torch.float64).contiguous()
    res = {}
    for name, flags in (("plain", 0), ("merged", ops.FLAG_MERGE_SRC)):
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        grad = torch.zeros_like(table)
        if form == "scatter":
            ops.model_train_backward(table, trip, gd, b, loss, model, "riem", None, None, None, None, 1.0, 1.0, grad_table=grad, flags=flags)
        else:
            rows = torch.zeros(2 * b, 2, n, n, dtype=torch.float64, device=dev)
            wp = torch.zeros((b + 63) // 64, 2 + n, dtype=torch.float64, device=dev)
            ops.model_train_backward(table, trip, gd, b, loss, model, "riem", None, None, None, None, 1.0, 1.0, grad_rows=rows,
                                     wave_partials=wp, flags=flags)
            order, rowptr = ops.sorted_slots(torch.cat((trip[:, 0], trip[:, 1])), nodes, merged_src=b if flags else 0)
            ops.segment_sum_rows_(grad, rows, order, rowptr, wave_partials=wp, num_waves=(b + 63) // 64, partial_stride=2 + n, loss=loss)
        with pytest.raises(IndexError):
            ops.check_status(dev)
        res[name] = (grad.cpu(), loss.cpu())
    assert torch.isfinite(res["merged"][0]).all()
    assert float((res["merged"][0] - res["plain"][0]).abs().max()) <= 1e-12 * float(res["plain"][0].abs().max())
    assert abs(float(res["merged"][1] - res["plain"][1])) <= 1e-12 * abs(float(res["plain"][1]))
