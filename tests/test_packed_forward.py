"""GPU (`-m gpu`): the INDEXED forward over the packed table, dims 5..8 (C-ABI sympa_table_pack / sympa_model_forward_packed /
sympa_model_forward_batches_packed; csrc/siegel_packed_kernel.hpp) -- what Model.forward / forward_batches / evaluate run while the
table does not change between batches (sympa/model.py:16-30, sympa/embeddings.py:29-34, sympa/runner.py:124-135,142-154).

Parity: the golden vectors of the imported reference n = 5..8 through the packed path (1e-9; the arithmetic differs from the dense
kernel's: products with the inverted factor instead of triangular solves), packed == dense to 1e-12 on the bench tables, the
oracle on seeded inputs, properties at configs[3]'s full size, the error contract (bad index, point outside the manifold), and the
pack's life cycle (torch version counter, storage, optimiser steps)."""
import numpy as np
import pytest
import torch

from oracle import siegel_oracle as so
from tests.helpers import GOLDEN, METRICS, MODELS, T, points, rel_err, sym, upper_points

pytestmark = pytest.mark.gpu
TOL = 1e-9
TOL_FAR_VS_REFERENCE = 1e-6   # see tests/test_hostsim_parity.py


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from sympa_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def packed_dist(z1, z2, model, metric, w=None, dev="cuda:0"):
    """dist(z1[i], z2[i]) through the packed INDEXED path: the two point lists become one table, pair i = (i, b + i)."""
    from sympa_amd import ops
    z1, z2 = T(z1), T(z2)
    b = z1.shape[0]
    table = torch.cat((z1, z2)).to(dev).contiguous()
    trip = torch.stack((torch.arange(b), torch.arange(b) + b), 1).to(dev)
    pk = ops.PackedTable(model).ensure(table)
    out = ops.model_forward_packed(pk, trip, metric, None if w is None else T(w).to(dev))
    ops.check_status(torch.device(dev))
    return out.cpu()


@pytest.mark.parametrize("n", [5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_packed_forward_golden_vectors_of_the_reference(dev, model, n):
    g = np.load(f"{GOLDEN}/dist_{model}_n{n}.npz")
    for case in g["case_names"]:
        z1, z2 = g[f"{case}__z1"], g[f"{case}__z2"]
        for metric in METRICS:
            got = packed_dist(z1, z2, model, metric, g["wsum_weights"])
            tol = TOL_FAR_VS_REFERENCE if case in ("far", "s1.0") else TOL
            assert rel_err(got, g[f"{case}__{metric}"], atol=1e-12) < tol, (model, n, case, metric)


@pytest.mark.parametrize("n", [5, 6, 7, 8])
@pytest.mark.parametrize("model", MODELS)
def test_packed_forward_against_oracle_and_dense_kernel(dev, model, n):
    """Seeded inputs at three scales, ragged batch (not a multiple of 64), every metric: the oracle to 1e-9, the dense kernel to
    1e-12."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(2000 + n)
    N, b = 301, 777
    for s in (1e-3, 0.3, 0.8):
        table = points(model, N, n, s, g)
        trip = torch.stack((torch.randint(0, N, (b,), generator=g), torch.randint(0, N, (b,), generator=g)), 1)
        tab_d, trip_d = table.to(dev), trip.to(dev)
        pk = ops.PackedTable(model).ensure(tab_d)
        for metric in METRICS:
            w = torch.linspace(-0.3, 1.2, n)
            got = ops.model_forward_packed(pk, trip_d, metric, w.to(dev)).cpu()
            dense = ops.model_forward(tab_d, trip_d, model, metric, w.to(dev)).cpu()
            want = so.model_forward(table, trip, model, metric, w)
            assert rel_err(got, want) < TOL, (model, n, s, metric)
            assert rel_err(got, dense, atol=1e-13) < 1e-12, (model, n, s, metric)
    ops.check_status(dev)


@pytest.mark.parametrize("model,n,N,b", [("upper", 8, 45500, 262144), ("bounded", 8, 5041, 65536), ("upper", 6, 5041, 65536)])
def test_packed_forward_full_size_equals_dense_and_properties(dev, model, n, N, b):
    """configs[3]'s full size (upper n = 8, 262 144 pairs of 45 500 rows) and two more shapes on the BENCH tables: packed == dense
    to 1e-12, symmetry d(x, y) = d(y, x), d(x, x) = 0 exactly, a 256-pair oracle sample."""
    from sympa_amd import data, ops
    table = data.trained_like_table(N, n, model=model, seed=42).to(dev)
    trip = data.sample_pairs(N, b, 0, 42).to(dev)
    pk = ops.PackedTable(model).ensure(table)
    d_p = ops.model_forward_packed(pk, trip, "riem")
    d_d = ops.model_forward(table, trip, model, "riem")
    ops.check_status(dev)
    assert torch.isfinite(d_p).all() and (d_p > 0).all()
    assert rel_err(d_p.cpu(), d_d.cpu(), atol=1e-13) < 1e-12
    flipped = trip[:, [1, 0, 2]].contiguous() if trip.shape[1] > 2 else trip.flip(1).contiguous()
    assert rel_err(ops.model_forward_packed(pk, flipped, "riem").cpu(), d_p.cpu()) < 1e-10
    same = torch.stack((trip[:, 0], trip[:, 0]), 1).contiguous()
    assert torch.all(ops.model_forward_packed(pk, same, "riem") == 0)
    k = 256
    want = so.model_forward(table.cpu(), trip[:k].cpu(), model, "riem")
    assert rel_err(d_p[:k].cpu(), want) < 1e-8


def test_packed_forward_error_contract_and_edge_cases(dev):
    """Empty batch, one pair, strided triplets, index out of range (NaN + IndexError like the dense path), a point outside the
    manifold (reported by every pair it enters: AssertionError like siegel_manifold.py:64-66; not by the pack), non-finite input."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(9)
    table = upper_points(70, 6, 0.3, g).to(dev)
    pk = ops.PackedTable("upper").ensure(table)
    assert ops.model_forward_packed(pk, torch.zeros(0, 2, dtype=torch.int64, device=dev)).shape == (0,)
    one = ops.model_forward_packed(pk, torch.tensor([[1, 2]], device=dev))
    assert one.shape == (1,) and torch.isfinite(one).all()
    trip3 = torch.randint(0, 70, (200, 3), generator=g).to(dev)                # [b, 3] triplets: stride 3
    assert torch.equal(ops.model_forward_packed(pk, trip3), ops.model_forward_packed(pk, trip3[:, :2].contiguous()))
    ops.check_status(dev)
    bad = ops.model_forward_packed(pk, torch.tensor([[1, 2], [3, 70], [-1, 0]], device=dev))
    assert torch.isfinite(bad[0]) and torch.isnan(bad[1]) and torch.isnan(bad[2])
    with pytest.raises(IndexError):
        ops.check_status(dev)
    off = table.clone()
    off[4, 1] = -off[4, 1]
    pk_off = ops.PackedTable("upper").ensure(off)
    ops.check_status(dev)                                                      # packing alone raises nothing (nor does the reference)
    assert torch.isfinite(ops.model_forward_packed(pk_off, torch.tensor([[1, 2], [5, 7]], device=dev))).all()
    ops.check_status(dev)                                                      # ... nor do pairs that do not touch the point
    out = ops.model_forward_packed(pk_off, torch.tensor([[4, 5], [1, 2], [7, 4]], device=dev))
    assert torch.isnan(out[0]) and torch.isfinite(out[1]) and torch.isnan(out[2])
    with pytest.raises(AssertionError):                                        # every pair it enters does
        ops.check_status(dev)
    nanny = table.clone()
    nanny[9, 0, 2, 3] = float("nan")
    out = ops.model_forward_packed(ops.PackedTable("upper").ensure(nanny), torch.tensor([[9, 1], [1, 2]], device=dev))
    assert torch.isnan(out[0]) and torch.isfinite(out[1])
    with pytest.raises(AssertionError):
        ops.check_status(dev)
    with pytest.raises(ValueError):                                            # dims outside 5..8: no packed path
        ops.PackedTable("upper").ensure(upper_points(10, 4, 0.3, g).to(dev))


@pytest.mark.parametrize("model,n", [("upper", 8), ("bounded", 7), ("upper", 5)])
def test_packed_batches_equal_single_calls_bit_for_bit(dev, model, n):
    """sympa_model_forward_batches_packed: 40 ragged batches (more than one launch group of 32, an empty batch among them) ==
    one sympa_model_forward_packed call per batch, bit for bit (the per-pair arithmetic does not depend on the batch)."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(31)
    N = 400
    table = points(model, N, n, 0.4, g).to(dev)
    sizes = [int(x) for x in torch.randint(1, 900, (40,), generator=g)]
    sizes[7] = 0
    batches = [torch.randint(0, N, (b, 3), generator=g).to(dev) for b in sizes]
    outs = [torch.full((b,), -1.0, dtype=torch.float64, device=dev) for b in sizes]
    pk = ops.PackedTable(model)
    plan = ops.PackedBatchedForward(pk, table, batches, outs, "fone")
    plan.run()
    ops.check_status(dev)
    for t, o in zip(batches, outs):
        if t.shape[0]:
            assert torch.equal(o, ops.model_forward_packed(pk, t, "fone"))


def test_model_uses_the_pack_and_follows_the_table(dev):
    """Model.forward under no_grad takes the packed path from the SECOND call on an unchanged table; forward_batches / evaluate
    pack at once; an optimiser step (raw-pointer kernels: sympa_amd bumps the torch version counter), an in-place torch op, a
    replaced storage (`embeds.data = ...`) and load_state_dict all make the next call repack; `use_packed = False` and autograd
    calls stay on the dense kernel."""
    from sympa_amd import data, ops
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianSGD

    class A:
        manifold, metric, dims, num_points = "upper", "riem", 6, 500
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.trained_like_table(500, 6, model="upper", seed=3)
    m = m.to(dev)
    g = torch.Generator().manual_seed(1)
    trip = torch.randint(0, 500, (8192, 3), generator=g).to(dev)

    def dense():
        return ops.model_forward(m.embeddings.embeds.data, trip, "upper", "riem", None, m.scale.data, m.scale_coef)

    pk = m.packed_table()
    assert pk is not None and pk.repacks == 0
    with torch.no_grad():
        # default policy (round 6): with the device-side validity check on, single calls of the UPPER model stay on the dense kernel
        # (the check costs what the pack saves there); the list forms use the pack
        m(trip); m(trip); m(trip)
        assert pk.strict and pk.repacks == 0 and pk.pack is None
    pk.strict = False                                # the life cycle of the host key (rounds 5): single calls use the pack as well
    with torch.no_grad():
        a = m(trip)
        assert pk.repacks == 0                       # first sight of this table version: dense kernel
        b = m(trip)
        assert pk.repacks == 1                       # second: packed
        assert rel_err(b.cpu(), a.cpu(), atol=1e-13) < 1e-12
        m(trip)
        assert pk.repacks == 1
    # an optimiser step writes the table through raw pointers
    opt = RiemannianSGD(m.parameters(), lr=0.05, weight_decay=0.0, stabilize=None)
    m.embeddings.embeds.grad = torch.randn(m.embeddings.embeds.shape, generator=g, dtype=torch.float64).to(dev) * 0.1
    opt.step()
    with torch.no_grad():
        outs = m.forward_batches([trip])             # the list form packs at once
        assert pk.repacks == 2
        assert rel_err(outs[0].cpu(), dense().cpu(), atol=1e-13) < 1e-12
        assert not torch.allclose(outs[0], a)
        # an in-place torch op on the parameter
        m.embeddings.embeds.mul_(1.0)
        m.embeddings.embeds[:, 0].add_(0.01 * sym(torch.randn(500, 6, 6, generator=g, dtype=torch.float64)).to(dev))
        dd = m.evaluate(trip[:, :2].contiguous(), torch.ones(8192, dtype=torch.float64, device=dev), 2048)
        assert pk.repacks == 3
        # a new storage behind the same Parameter
        m.embeddings.embeds.data = data.trained_like_table(500, 6, model="upper", seed=4).to(dev)
        c = m.forward_batches([trip])[0].clone()
        assert pk.repacks == 4
        assert rel_err(c.cpu(), dense().cpu(), atol=1e-13) < 1e-12
        # load_state_dict copies in place
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["embeddings.embeds"] = data.trained_like_table(500, 6, model="upper", seed=5).to(dev)
        m.load_state_dict(sd)
        e = m.forward_batches([trip])[0]
        assert pk.repacks == 5
        assert rel_err(e.cpu(), dense().cpu(), atol=1e-13) < 1e-12
        assert dd > 0
        m.use_packed = False
        assert m.packed_table() is None
        assert torch.equal(m(trip), dense())
    ops.check_status(dev)


# ---- spd (configs[4]; PARITY UNPINNED like every spd entry: geoopt is absent from the reference tree) ----------------------------

def spd_packed_dist(x, y, dev="cuda:0"):
    from sympa_amd import ops
    b = x.shape[0]
    table = torch.cat((x, y)).to(dev).contiguous()
    trip = torch.stack((torch.arange(b), torch.arange(b) + b), 1).to(dev)
    pk = ops.SpdPackedTable().ensure(table)
    out = ops.spd_model_forward_packed(pk, trip)
    ops.check_status(torch.device(dev))
    return out.cpu()


@pytest.mark.parametrize("n", [8, 16])
def test_spd_packed_forward_against_mpmath_goldens(dev, monkeypatch, n):
    """The 50-digit evaluations of the published formula (tests/golden/spd_n*.npz) through the packed path: the tolerances of
    tests/test_spd.py::test_gpu_spd_against_mpmath_goldens (the factor is the one the dense kernel computes)."""
    from sympa_amd import ops
    monkeypatch.setattr(ops, "SPD_PACKED_DIMS", frozenset(range(6, 17)))
    g = np.load(f"{GOLDEN}/spd_n{n}.npz")
    for case in g["case_names"]:
        x, y, want = torch.from_numpy(g[f"{case}__x"]), torch.from_numpy(g[f"{case}__y"]), g[f"{case}__dist_exact50"]
        got = spd_packed_dist(x, y)
        tol = {"s1.5": 2e-7 if n < 16 else 3e-6, "cond1e6": 1e-9}.get(str(case), 1e-11)
        assert rel_err(got, want, atol=1e-13) < tol, (n, case)
        if case == "same":
            assert torch.all(got == 0)


@pytest.mark.parametrize("n", list(range(6, 17)))
def test_spd_packed_forward_equals_dense_kernel_every_size(dev, monkeypatch, n):
    """Every instantiation (n = 6..16; the binding uses the packed path at n = 16 only, where it is faster): packed == dense to
    1e-12 and the oracle to 1e-10 on seeded tables of three scales, ragged batch, bad index, a point that is not positive definite."""
    from sympa_amd import ops
    from tests.helpers import spd_points
    monkeypatch.setattr(ops, "SPD_PACKED_DIMS", frozenset(range(6, 17)))
    g = torch.Generator().manual_seed(3000 + n)
    N, b = 211, 333
    for s in (1e-3, 0.3, 0.8):
        table = spd_points(N, n, s, g)
        trip = torch.stack((torch.randint(0, N, (b,), generator=g), torch.randint(0, N, (b,), generator=g)), 1)
        tab_d, trip_d = table.to(dev), trip.to(dev)
        pk = ops.SpdPackedTable().ensure(tab_d)
        got = ops.spd_model_forward_packed(pk, trip_d).cpu()
        dense = ops.spd_model_forward(tab_d, trip_d).cpu()
        assert rel_err(got, dense, atol=1e-13) < 1e-12, (n, s)
        assert rel_err(got, so.spd_dist(table[trip[:, 0]], table[trip[:, 1]]), atol=1e-12) < 1e-10, (n, s)
    ops.check_status(dev)
    bad = ops.spd_model_forward_packed(pk, torch.tensor([[1, 2], [3, N], [-1, 0]], device=dev))
    assert torch.isfinite(bad[0]) and torch.isnan(bad[1]) and torch.isnan(bad[2])
    with pytest.raises(IndexError):
        ops.check_status(dev)
    off = tab_d.clone()
    off[4] = -off[4]
    pk_off = ops.SpdPackedTable().ensure(off)
    ops.check_status(dev)                          # packing alone raises nothing
    out = ops.spd_model_forward_packed(pk_off, torch.tensor([[4, 5], [1, 2]], device=dev))
    assert torch.isnan(out[0]) and torch.isfinite(out[1])
    with pytest.raises(AssertionError):
        ops.check_status(dev)


def test_spd_packed_forward_full_size_and_model(dev):
    """configs[4] at full size (n = 16, 100 000 points, 1 048 576 pairs): packed == dense to 1e-12, symmetry, d(x, x) = 0; and the
    spd Model takes the packed path under no_grad from the second call on an unchanged table and repacks after an optimiser step."""
    from sympa_amd import data, ops
    from sympa_amd.model import Model
    from sympa_amd.optim import RiemannianSGD
    n, rows, b = 16, 100_000, 1 << 20
    table = data.spd_table(rows, n, scale=0.1, seed=7).to(dev)
    trip = data.sample_pairs(rows, b, 0, 7).to(dev)
    pk = ops.SpdPackedTable().ensure(table)
    d_p = ops.spd_model_forward_packed(pk, trip)
    d_d = ops.spd_model_forward(table, trip)
    ops.check_status(dev)
    assert torch.isfinite(d_p).all() and (d_p > 0).all()
    assert rel_err(d_p.cpu(), d_d.cpu(), atol=1e-13) < 1e-12
    flipped = trip[:, [1, 0]].contiguous()
    assert rel_err(ops.spd_model_forward_packed(pk, flipped).cpu(), d_p.cpu()) < 1e-11
    same = torch.stack((trip[:, 0], trip[:, 0]), 1).contiguous()
    assert torch.all(ops.spd_model_forward_packed(pk, same) == 0)
    del table, pk, d_p, d_d

    class A:
        manifold, metric, dims, num_points = "spd", "riem", 16, 2000
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.spd_table(2000, 16, scale=0.3, seed=3)
    m = m.to(dev)
    g = torch.Generator().manual_seed(1)
    t = torch.randint(0, 2000, (8192, 3), generator=g).to(dev)
    pk = m.packed_table()
    assert pk is not None and pk.repacks == 0
    with torch.no_grad():
        a = m(t)
        assert pk.repacks == 0
        b2 = m(t)
        assert pk.repacks == 1 and rel_err(b2.cpu(), a.cpu(), atol=1e-13) < 1e-12
    opt = RiemannianSGD(m.parameters(), lr=0.01, weight_decay=0.0, stabilize=None)
    gr = torch.randn(m.embeddings.embeds.shape, generator=g, dtype=torch.float64).to(dev) * 0.05
    m.embeddings.embeds.grad = 0.5 * (gr + gr.transpose(-1, -2))
    opt.step()
    with torch.no_grad():
        c = m.forward_batches([t])[0]
        assert pk.repacks == 2
        assert rel_err(c.cpu(), ops.spd_model_forward(m.embeddings.embeds.data, t, m.scale.data, m.scale_coef).cpu(), atol=1e-13) < 1e-12
        assert not torch.allclose(c, a)
    ops.check_status(dev)


def test_bounded_model_packs_at_first_sight_on_large_calls(dev):
    """The bounded model's dense kernel factors I - W W^H per pair; from 4 pairs per table row on, one pack + the packed kernel is
    the faster route even for a single call: Model.forward packs at FIRST sight there (small calls still wait for the second)."""
    from sympa_amd import data, ops
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "bounded", "finf", 7, 600
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.trained_like_table(600, 7, model="bounded", seed=3)
    m = m.to(dev)
    g = torch.Generator().manual_seed(1)
    small = torch.randint(0, 600, (2000, 3), generator=g).to(dev)          # < 4 N pairs (and < PACKED_MIN_PAIRS)
    big = torch.randint(0, 600, (8192, 3), generator=g).to(dev)            # >= 4 N
    pk = m.packed_table()
    with torch.no_grad():
        m(small)
        assert pk.repacks == 0
        got = m(big)
        assert pk.repacks == 1
        want = ops.model_forward(m.embeddings.embeds.data, big, "bounded", "finf", None, m.scale.data, m.scale_coef)
        assert rel_err(got.cpu(), want.cpu(), atol=1e-13) < 1e-12
    ops.check_status(dev)


# ---- round 6: validity of the pack decided on the device (C-ABI sympa_table_digest / sympa_table_pack_refresh) -------------------

def test_table_digest_sees_every_change(dev):
    """sympa_table_digest: `changed` is 1 on the first call over a zeroed state, 0 while the bytes stay, 1 after ANY word changed
    (first, last, a single low bit, two words swapped), 1 with SYMPA_FLAG_DIGEST_FORCE; odd word counts; the change counter."""
    import ctypes
    from sympa_amd import _lib
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def digest(buf, state, flags=0):
        _lib.check(lib.sympa_table_digest(buf.data_ptr(), buf.numel() * 8, state.data_ptr(), flags, stream))
        return int(state.view(torch.int32)[6])

    g = torch.Generator().manual_seed(5)
    for words in (2, 3, 777, 1 << 20, (1 << 22) + 1):
        buf = torch.randn(words, generator=g, dtype=torch.float64).to(dev)
        state = torch.zeros(4096, dtype=torch.uint8, device=dev)
        assert digest(buf, state) == 1
        assert digest(buf, state) == 0 and digest(buf, state) == 0
        for pos in (0, words - 1, words // 2):
            bits = buf.view(torch.int64)
            bits[pos] ^= 1                                    # one ulp of one word
            assert digest(buf, state) == 1, (words, pos)
            assert digest(buf, state) == 0
        if words >= 3:
            a, b = buf[0].clone(), buf[words - 1].clone()
            buf[0], buf[words - 1] = b, a                     # same multiset of words, different places
            assert digest(buf, state) == 1
        assert digest(buf, state, flags=1) == 1               # SYMPA_FLAG_DIGEST_FORCE
        assert digest(buf, state) == 0
        assert int(state.view(torch.int32)[7]) == (6 if words >= 3 else 5)
        assert int(state.view(torch.int32)[4]) == 0          # the ticket counter is back at 0


@pytest.mark.parametrize("kind", ["upper8", "bounded7", "spd16"])
def test_data_writes_that_move_no_version_counter_are_seen(dev, kind):
    """The reference's era writes tables through `.data` (embeddings.py:36-39; torch-1.5 / geoopt optimisers `p.data.add_()`): no
    torch version counter moves.  Between two no_grad forwards such a write must give the NEW distances (against the oracle)
    without any invalidate(); the device found the change by itself (`device_repacks`), the host key never moved (`repacks`)."""
    from sympa_amd import data, ops
    from sympa_amd.model import Model
    name, n = {"upper8": ("upper", 8), "bounded7": ("bounded", 7), "spd16": ("spd", 16)}[kind]

    class A:
        manifold, metric, dims, num_points = name, "riem", n, 700
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = (data.spd_table(700, n, scale=0.3, seed=3) if name == "spd"
                                    else data.trained_like_table(700, n, model=name, seed=3))
    m = m.to(dev)
    g = torch.Generator().manual_seed(2)
    trip = torch.randint(0, 700, (8192, 3), generator=g).to(dev)

    def oracle_now():
        tab = m.embeddings.embeds.data.cpu()
        sel = trip[:256].cpu()
        if name == "spd":
            return so.spd_dist(tab[sel[:, 0]], tab[sel[:, 1]])
        return so.model_forward(tab, sel, name, "riem", scale=m.scale.data.cpu(), scale_coef=1.0)

    pk = m.packed_table()
    assert pk is not None and pk.strict
    # (single calls of the upper model stay dense under the strict check: its packed path is the list form)
    fwd = (lambda: m.forward_batches([trip])[0]) if name == "upper" else (lambda: m(trip))
    with torch.no_grad():
        fwd(); a = fwd().clone()                          # (second single call: packed; the list form packs at once)
        assert pk.repacks == 1 and pk.device_repacks() == 1
        assert rel_err(a[:256].cpu(), oracle_now()) < 1e-8
        fwd()
        assert pk.device_repacks() == 1                   # unchanged table: digest equal, the pack kernel returned at once
        v = m.embeddings.embeds._version
        # (NOT a scaling of the whole point: Z -> c Z and X -> c X are isometries, the distances would not move)
        if name == "bounded":
            m.embeddings.embeds.data.mul_(0.97)           # stays inside the domain
        elif name == "upper":
            m.embeddings.embeds.data[:, 1].mul_(1.05)     # Y stays positive definite
        else:
            m.embeddings.embeds.data.add_(0.05 * torch.eye(n, dtype=torch.float64, device=dev))
        assert m.embeddings.embeds._version == v          # ... and torch saw nothing
        b = fwd().clone()
        assert pk.repacks == 1 and pk.device_repacks() == 2
        assert rel_err(b[:256].cpu(), oracle_now()) < 1e-8
        assert not torch.allclose(a, b)
        # the list forms check too
        m.embeddings.embeds.data[5].mul_(1.0 + 1e-9)      # one row, nine digits down
        c = m.forward_batches([trip])[0].clone()
        assert pk.device_repacks() == 3
        if name == "spd":
            want = ops.spd_model_forward(m.embeddings.embeds.data, trip, m.scale.data, m.scale_coef)
        else:
            want = ops.model_forward(m.embeddings.embeds.data, trip, name, "riem", None, m.scale.data, m.scale_coef)
        assert rel_err(c.cpu(), want.cpu(), atol=1e-13) < 1e-11
        # strict off: the key alone is trusted (the documented round-5 behaviour) -- the write is NOT seen
        pk.strict = False
        if name == "spd":
            m.embeddings.embeds.data.add_(0.05 * torch.eye(n, dtype=torch.float64, device=dev))
        else:
            m.embeddings.embeds.data[:, 1].mul_(0.99 if name == "bounded" else 1.02)
        stale = fwd().clone()
        assert torch.equal(stale, c) and pk.device_repacks() == 3
        pk.strict = True
        fresh = fwd()
        assert pk.device_repacks() == 4 and not torch.allclose(fresh, c)
    ops.check_status(dev)


def test_captured_forward_repacks_by_itself(dev):
    """Round-5 advice: a forward captured into a hipGraph while the pack was current recorded only the pair kernel; replays after
    an optimiser step read a stale pack.  Now the capture records digest + guarded pack + pair kernel: a replay after the table
    changed (through torch, through `.data`, through the raw-pointer optimiser) gives the new distances.  A capture BEFORE any pack
    exists allocates nothing into the graph's pool: it records the dense kernel.  (Bounded model: the one whose single calls
    use the pack under the strict check.)"""
    from sympa_amd import data, ops
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "bounded", "fone", 8, 600
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    with torch.no_grad():
        m.embeddings.embeds.data = data.trained_like_table(600, 8, model="bounded", seed=3)
    m = m.to(dev)
    g = torch.Generator().manual_seed(2)
    trip = torch.randint(0, 600, (8192, 3), generator=g).to(dev)

    def dense():
        return ops.model_forward(m.embeddings.embeds.data, trip, "bounded", "fone", None, m.scale.data, m.scale_coef)

    pk = m.packed_table()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.no_grad(), torch.cuda.stream(side):
        # (1) no pack yet: the capture must not create one
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=side):
            out0 = m(trip)
            out0b = m(trip)           # "second sight" inside the capture: still no allocation into the pool
        assert pk.pack is None and pk.repacks == 0
        g0.replay()
        side.synchronize()
        assert torch.equal(out0, dense()) and torch.equal(out0b, out0)
        # (2) with a pack: digest + guarded pack + pair kernel are recorded
        m(trip); m(trip)
        assert pk.repacks == 1
        key = pk.key
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=side):
            out1 = m(trip)
        assert pk.key == key          # a capture runs nothing: the host key is left alone
        g1.replay()
        side.synchronize()
        assert rel_err(out1.cpu(), dense().cpu(), atol=1e-13) < 1e-12
        before = out1.clone()
        m.embeddings.embeds.data.mul_(0.97)              # invisible to torch (stays inside the domain; not an isometry)
        g1.replay()
        side.synchronize()
        assert rel_err(out1.cpu(), dense().cpu(), atol=1e-13) < 1e-12 and not torch.allclose(out1, before)
        m.embeddings.embeds.mul_(0.98)                   # visible to torch: the recorded launches do not care either way
        g1.replay()
        side.synchronize()
        assert rel_err(out1.cpu(), dense().cpu(), atol=1e-13) < 1e-12
        # eager call afterwards: the key moved -> unconditional repack, same values
        assert rel_err(m(trip).cpu(), dense().cpu(), atol=1e-13) < 1e-12
    torch.cuda.current_stream(dev).wait_stream(side)
    ops.check_status(dev)


def test_pack_written_on_one_stream_read_on_another(dev):
    """Round-5 advice: the pack is written on the stream current in ensure() and may be read from another one: the reader is
    ordered behind the writer's stream (an event recorded at the switch)."""
    from sympa_amd import ops
    g = torch.Generator().manual_seed(8)
    table = points("upper", 3000, 8, 0.4, g).to(dev)
    trip = torch.randint(0, 3000, (65536, 2), generator=g).to(dev)
    want = ops.model_forward(table, trip, "upper", "riem")
    torch.cuda.synchronize(dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    pk = ops.PackedTable("upper")
    for _ in range(5):
        with torch.cuda.stream(s1):
            table.mul_(1.0)                       # version moves: forced repack on s1
            torch.cuda._sleep(20_000_000)         # ... which starts late
            pk.ensure(table)
        with torch.cuda.stream(s2):
            got = ops.model_forward_packed(pk.ensure(table), trip, "riem")
        s2.synchronize()
        assert rel_err(got.cpu(), want.cpu(), atol=1e-13) < 1e-12
    torch.cuda.synchronize(dev)
    ops.check_status(dev)


def test_spd_packed_entry_falls_back_when_the_instantiation_is_demoted(dev):
    """Round-5 advice: with the sixteen-lanes spd forward demoted by the self-check the packed C entry used to fail; it now runs
    the one-lane kernel over the pack's images, Model.packed_table() stops offering the pack, and the self-check covers the
    PACKED instantiation (a separate inline-asm binary) for every n = 6..16."""
    from sympa_amd import _lib, data, ops, selfcheck
    from sympa_amd.model import Model
    lib = _lib.load()
    table = data.spd_table(300, 16, scale=0.3, seed=3).to(dev)
    g = torch.Generator().manual_seed(3)
    trip = torch.randint(0, 300, (5000, 2), generator=g).to(dev)
    pk = ops.SpdPackedTable().ensure(table)
    fast = ops.spd_model_forward_packed(pk, trip)
    _lib.check(lib.sympa_set_instance_fallback(selfcheck.SPD_FWD, 0, 16, 1))
    try:
        slow = ops.spd_model_forward_packed(pk, trip)        # C entry: one-lane kernel, rows of n (n + 1) doubles
        assert rel_err(slow.cpu(), fast.cpu(), atol=1e-13) < 1e-10
        assert not ops.SpdPackedTable.supported(table)

        class A:
            manifold, metric, dims, num_points = "spd", "riem", 16, 300
            scale_coef, scale_init, train_scale = 1.0, 1.0, False

        assert Model(A).to(dev).packed_table() is None
    finally:
        _lib.check(lib.sympa_set_instance_fallback(selfcheck.SPD_FWD, 0, 16, 0))
    assert ops.SpdPackedTable.supported(table)
    # every packed instantiation passes the first-use comparison on this build
    for n in range(6, 17):
        assert all(ok for ok, _ in selfcheck._check_spd_fwd("spd", n, dev)), n
    ops.check_status(dev)
