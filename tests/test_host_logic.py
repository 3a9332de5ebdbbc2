"""CPU: host-side integer/index logic is bit-exact: keyed RNG, pair sampling, graph
triplets (vs networkx BFS), DistributedSampler sharding (vs torch's own sampler)."""
import numpy as np
import pytest
import torch
from torch.utils.data import DistributedSampler, TensorDataset

from sympa_amd import data


def _splitmix_scalar(x):
    m = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def test_keyed_rng_matches_scalar_python():
    seed, stream = 42, 10
    key = _splitmix_scalar(((seed << 32) | stream) & ((1 << 64) - 1))
    want = [_splitmix_scalar(c ^ key) for c in range(50)]
    got = data.keyed_u64(seed, stream, np.arange(50)).tolist()
    assert got == want


def test_sample_pairs_bit_exact_and_rank_independent():
    n, b = 1093, 4096
    p = data.sample_pairs(n, b, batch_id=3, seed=42)
    assert p.dtype == torch.int64 and p.shape == (b, 2)
    assert (p[:, 0] != p[:, 1]).all() and p.min() >= 0 and p.max() < n
    # pure-python restatement of the same keyed draw
    m = (1 << 64) - 1
    k10 = _splitmix_scalar(((42 << 32) | 10) & m)
    k11 = _splitmix_scalar(((42 << 32) | 11) & m)
    for k in (0, 1, 77, b - 1):
        c = 3 * b + k
        i = _splitmix_scalar(c ^ k10) % n
        j = (i + 1 + _splitmix_scalar(c ^ k11) % (n - 1)) % n
        assert p[k].tolist() == [i, j]
    # same bytes whoever generates them
    assert torch.equal(p, data.sample_pairs(n, b, batch_id=3, seed=42))


@pytest.mark.parametrize("name,nodes,triplets,diam", [("grid3d-125", 125, 7750, 12)])
def test_graph_triplets_match_networkx_bfs(name, nodes, triplets, diam):
    import networkx as nx
    g = data.named_graph(name)
    trip, id2node = data.graph_triplets(g)
    assert len(id2node) == nodes and trip.shape == (triplets, 3) and int(trip[:, 2].max()) == diam
    relabelled = nx.convert_node_labels_to_integers(g, ordering="sorted")
    sp = dict(nx.all_pairs_shortest_path_length(relabelled))
    want = [(i, j, sp[i][j]) for i in range(nodes) for j in range(i + 1, nodes)]
    assert trip.tolist() == [list(t) for t in want]          # lexicographic, exact integers


def test_graph_triplets_tree_b3_h6_match_networkx_bfs():
    """configs[1]: balanced tree b = 3, h = 6 -- every one of the 596 778 triplets against a networkx BFS per source."""
    import networkx as nx
    g = data.named_graph("tree-b3-h6")
    trip, id2node = data.graph_triplets(g)
    assert len(id2node) == 1093 and trip.shape == (1093 * 1092 // 2, 3) and int(trip[:, 2].max()) == 12
    relabelled = nx.convert_node_labels_to_integers(g, ordering="sorted")
    want = []
    for i in range(1093):
        sp = nx.single_source_shortest_path_length(relabelled, i)
        want.extend((i, j, sp[j]) for j in range(i + 1, 1093))
    assert torch.equal(trip, torch.tensor(want, dtype=torch.int64))


def test_graph_triplets_margulis_71_multigraph_with_self_loops():
    """configs[2]: margulis_gabber_galil_graph(71) is a MultiGraph on 71^2 = 5 041 nodes with parallel edges and 284
    self-loops (preprocess.py:60-61) -- the case where an adjacency-matrix BFS can go wrong (loops on the diagonal,
    multiplicities as weights).  All 12.7 M triplets are generated; the rows of 150 source nodes (every 40th, plus
    nodes that carry self-loops, plus the last ones) are compared with a networkx BFS, the rest through invariants."""
    import networkx as nx
    g = data.named_graph("margulis-71")
    assert g.is_multigraph() and nx.number_of_selfloops(g) == 284
    trip, id2node = data.graph_triplets(g)
    n = 5041
    assert len(id2node) == n and trip.shape == (n * (n - 1) // 2, 3)          # connected: every pair has a distance
    assert id2node[0] == (0, 0) and id2node[n - 1] == (70, 70)                 # sorted() relabelling of (x, y) nodes
    # lexicographic order of (i, j), i < j
    iu, ju = torch.triu_indices(n, n, offset=1)
    assert torch.equal(trip[:, 0], iu) and torch.equal(trip[:, 1], ju)
    assert int(trip[:, 2].min()) == 1
    relabelled = nx.convert_node_labels_to_integers(g, ordering="sorted")
    loops = sorted({u for u, _ in nx.selfloop_edges(relabelled)})
    sources = sorted(set(range(0, n, 40)) | set(loops[:12]) | {n - 3, n - 2})
    starts = torch.cumsum(torch.tensor([0] + [n - 1 - i for i in range(n - 1)]), 0)     # first row of source i
    diam = 0
    for i in sources:
        sp = nx.single_source_shortest_path_length(relabelled, i)
        want = torch.tensor([sp[j] for j in range(i + 1, n)], dtype=torch.int64)
        got = trip[int(starts[i]):int(starts[i]) + n - 1 - i, 2]
        assert torch.equal(got, want), i
        diam = max(diam, max(sp.values()))
    assert int(trip[:, 2].max()) >= diam
    # every edge of the simple graph is a triplet at distance 1, and nothing else is
    simple = {(min(u, v), max(u, v)) for u, v in relabelled.edges() if u != v}
    one = trip[trip[:, 2] == 1]
    assert {(int(a), int(b)) for a, b in one[:, :2].tolist()} == simple


@pytest.mark.parametrize("length,world", [(103, 4), (8, 3), (5, 8), (64, 2), (7750, 8)])
@pytest.mark.parametrize("drop_last", [False, True])
def test_distributed_sampler_semantics(length, world, drop_last):
    ds = TensorDataset(torch.arange(length))
    for rank in range(world):
        s = DistributedSampler(ds, num_replicas=world, rank=rank, seed=3, drop_last=drop_last)
        s.set_epoch(2)
        assert list(s) == data.distributed_sampler_indices(length, world, rank, epoch=2, seed=3,
                                                           drop_last=drop_last)


def test_tables_are_on_the_manifold():
    t = data.trained_like_table(64, 4)
    assert t.shape == (64, 2, 4, 4) and t.dtype == torch.float64
    assert torch.equal(t, t.transpose(-1, -2))
    assert (torch.linalg.eigvalsh(t[:, 1]) > 0).all()
    w = data.trained_like_table(64, 4, model="bounded")
    wc = torch.complex(w[:, 0], w[:, 1])
    assert (torch.linalg.svdvals(wc) < 1).all()
    i = data.init_table(64, 3)
    assert ((i[:, 1] - torch.eye(3)).abs() <= 1e-3).all() and (i[:, 0].abs() <= 1e-3).all()


def test_preprocessed_file_round_trip_in_the_reference_format(tmp_path):
    """preprocess.py:165-171 writes {"triplets": set of (i, j, d), "id2node": dict}; train.py:80-97 reads it."""
    g = data.named_graph("grid3d-125")
    trip, id2node = data.graph_triplets(g)
    path = tmp_path / "preprocessed-data.pt"
    data.save_preprocessed(path, trip, id2node)
    raw = torch.load(path, weights_only=False)           # what the reference's train.py would see
    assert isinstance(raw["triplets"], set) and len(raw["triplets"]) == 7750
    assert all(isinstance(t, tuple) and len(t) == 3 and isinstance(t[2], int) for t in raw["triplets"])
    assert raw["id2node"][0] == (0, 0, 0) and len(raw["id2node"]) == 125
    ids, dist, id2 = data.load_preprocessed(path)
    assert torch.equal(ids, trip[:, :2]) and torch.equal(dist, trip[:, 2].to(torch.float64)) and id2 == id2node
    scaled = data.scale_triplet_distances(dist)
    assert float(scaled.max()) == 1.0 and float(scaled.min()) == pytest.approx(1 / 144)


def test_checkpoint_round_trip_with_ddp_prefix(tmp_path):
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "upper", "wsum", 3, 12
        scale_coef, scale_init, train_scale = 1.0, 1.0, True

    m1, m2 = Model(A), Model(A)
    path = tmp_path / "ckpt"
    data.save_checkpoint(path, m1, {i: i for i in range(12)})
    blob = torch.load(path, weights_only=False)
    assert set(blob["model"]) == {"module.scale", "module.embeddings.embeds", "module.manifold.metric.weights",
                                  "module.embeddings.manifold.metric.weights"}      # runner.py:160 layout
    assert data.load_checkpoint(path, m2) == {i: i for i in range(12)}
    assert torch.equal(m1.embeddings.embeds.data, m2.embeddings.embeds.data)


def test_check_all_points_batched_matches_per_point_loop():
    from sympa_amd.model import Model

    class A:
        manifold, metric, dims, num_points = "upper", "riem", 3, 20
        scale_coef, scale_init, train_scale = 1.0, 1.0, False

    m = Model(A)
    assert m.check_all_points() == (True, None, None)
    m.embeddings.embeds.data[7, 1] = -m.embeddings.embeds.data[7, 1]          # det(Y) < 0 for odd n
    ok, point, reason = m.check_all_points()
    assert not ok and torch.equal(point, m.embeddings.embeds.data[7]) and "determinant" in reason
    m.embeddings.embeds.data[3, 0, 0, 1] += 1.0                                # asymmetric row comes first
    ok, point, reason = m.check_all_points()
    assert not ok and torch.equal(point, m.embeddings.embeds.data[3]) and "symmetric" in reason


def test_manifold_parameter_survives_deepcopy_and_pickle():
    """copy.deepcopy(model) must keep the parameter on its manifold (nn.Parameter.__deepcopy__ would drop it and the
    optimiser would then update the copy with the Euclidean rule)."""
    import copy
    import pickle
    from sympa_amd.manifolds import UpperHalfManifold
    from sympa_amd.manifolds.base import ManifoldParameter
    man = UpperHalfManifold(dims=2)
    p = ManifoldParameter(torch.zeros(3, 2, 2, 2, dtype=torch.float64), manifold=man)
    q = copy.deepcopy(p)
    assert isinstance(q, ManifoldParameter) and isinstance(q.manifold, UpperHalfManifold) and q.requires_grad
    assert q.data_ptr() != p.data_ptr()
    r = pickle.loads(pickle.dumps(p))
    assert isinstance(r.manifold, UpperHalfManifold)


def test_dpp_hazard_checker_flags_a_copy_in_front_of_a_dpp_read():
    """tools/check_dpp_hazards.py (run by build() on every translation unit): a VALU write of a DPP source within two wait
    states is flagged -- directly, through a register pair, and across a branch into the block; wait states from s_nop
    and from unrelated instructions clear it."""
    import os
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_dpp_hazards as chk

    def scan(body):
        text = "_Z1kv:\n" + "".join("\t" + l + "\n" if not l.endswith(":") else l + "\n" for l in body) + "\ts_endpgm\n"
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            f.write(text)
        try:
            return chk.check_file(f.name)
        finally:
            os.unlink(f.name)

    dpp = "v_fmac_f64_dpp v[76:77], v[30:31], v[70:71] row_newbcast:0 row_mask:0xf bank_mask:0xf"
    assert scan(["v_mov_b64_e32 v[30:31], v[26:27]", dpp]) == (1, [("_Z1kv", "v_mov_b64_e32 v[30:31], v[26:27]", dpp)])
    assert len(scan(["v_accvgpr_read_b32 v31, a5", "v_add_f64 v[2:3], v[4:5], v[6:7]", dpp])[1]) == 1     # one wait state only
    assert scan(["v_mov_b64_e32 v[30:31], v[26:27]", "s_nop 1", dpp])[1] == []
    assert scan(["v_mov_b64_e32 v[30:31], v[26:27]", "v_add_f64 v[2:3], v[4:5], v[6:7]", "v_add_f64 v[8:9], v[4:5], v[6:7]", dpp])[1] == []
    assert scan(["v_mov_b64_e32 v[32:33], v[26:27]", dpp])[1] == []                                       # another register
    assert scan(["global_load_dwordx2 v[30:31], v[0:1], off", dpp])[1] == []                              # not a VALU write
    # the write sits at the end of a block that branches to the DPP instruction's block
    body = ["v_mov_b64_e32 v[30:31], v[26:27]", "s_cbranch_execz .LBB0_2", ".LBB0_1:", "s_nop 3", ".LBB0_2:", dpp]
    assert len(scan(body)[1]) == 1
    assert scan(["v_cmpx_lt_f64_e32 v[0:1], v[2:3]"])[1] != []
    # partial bank mask: the destination's old value is a DPP read too (two complementary masked writes back to back)
    half_a = "v_fmac_f64_dpp v[8:9], -v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0x3"
    half_b = "v_fmac_f64_dpp v[8:9], -v[2:3], v[4:5] row_newbcast:11 row_mask:0xf bank_mask:0xc"
    assert len(scan(["s_nop 1", half_a, half_b])[1]) == 1
    assert scan(["s_nop 1", half_a, "s_nop 1", half_b])[1] == []


def test_sorted_slots_lists_of_the_deterministic_gradient_accumulation():
    """ops.sorted_slots (pure torch, runs on any device): for every batch the slots 0 .. 2b-1 sorted by table row --
    stable, so inside a table row the slots keep batch order -- and CSR pointers of the table rows into that list."""
    import torch
    from sympa_amd import ops
    g = torch.Generator().manual_seed(3)
    steps, b, rows = 3, 50, 17
    src = torch.randint(0, rows, (steps, b), generator=g)
    dst = torch.randint(0, rows, (steps, b), generator=g)
    keys = torch.cat((src, dst), dim=1)
    order, rowptr = ops.sorted_slots(keys, rows)
    assert order.dtype == torch.int32 and rowptr.dtype == torch.int32
    assert order.shape == (steps, 2 * b) and rowptr.shape == (steps, rows + 1)
    for s in range(steps):
        assert int(rowptr[s, 0]) == 0 and int(rowptr[s, -1]) == 2 * b
        for r in range(rows):
            seg = order[s, int(rowptr[s, r]):int(rowptr[s, r + 1])].tolist()
            assert seg == [k for k in range(2 * b) if int(keys[s, k]) == r]      # exactly the slots of row r, in batch order
    o1, p1 = ops.sorted_slots(keys[0], rows)                                      # a single batch as a 1-d list
    assert torch.equal(o1[0], order[0]) and torch.equal(p1[0], rowptr[0])


def test_riemannian_adam_plain_parameters_match_torch_adam_and_state_snapshot():
    """sympa_amd.optim.RiemannianAdam keeps b1^t, b2^t in tensors advanced by the step (so that the step can be captured in a
    hipGraph); on parameters without a manifold it is the ordinary Adam: same trajectory as torch.optim.Adam.  snapshot_state /
    restore_state undo steps exactly (GraphedTrainStep's warm-up)."""
    import torch
    from sympa_amd.optim import RiemannianAdam
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(7, 3, generator=g, dtype=torch.float64)
    a = torch.nn.Parameter(w0.clone())
    b = torch.nn.Parameter(w0.clone())
    ours = RiemannianAdam([a], lr=0.05, eps=1e-7)
    ref = torch.optim.Adam([b], lr=0.05, eps=1e-7)
    ours.init_state()
    snap = ours.snapshot_state()
    a.grad = torch.ones_like(a)
    ours.step()
    ours.step()                         # two warm-up steps ...
    with torch.no_grad():
        a.copy_(w0)
    ours.restore_state(snap)            # ... leave no trace
    assert ours.param_groups[0]["step"] == 0 and float(ours.state[a]["bias_pows"][0]) == 1.0
    for it in range(6):
        grad = torch.randn(7, 3, generator=g, dtype=torch.float64)
        a.grad = grad.clone()
        b.grad = grad.clone()
        ours.step()
        ref.step()
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-12, atol=1e-14), it
    assert abs(float(ours.state[a]["bias_pows"][0]) - 0.9 ** 6) < 1e-15


def test_riemannian_adam_follows_changed_betas_and_loads_the_old_state_format():
    """Round-3 ADVICE: (i) geoopt evaluates betas ** step from the live group, so after group["betas"] changes the bias
    corrections must use the new betas like the moments do (same trajectory as torch.optim.Adam with the same change);
    (ii) a state dict written before the powers lived on the device (no bias_pows / betas entries) loads and steps."""
    import torch
    from sympa_amd.optim import RiemannianAdam
    g = torch.Generator().manual_seed(11)
    w0 = torch.randn(5, 2, generator=g, dtype=torch.float64)
    a, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    ours, ref = RiemannianAdam([a], lr=0.05, eps=1e-7), torch.optim.Adam([b], lr=0.05, eps=1e-7)
    for it in range(7):
        if it == 3:
            ours.param_groups[0]["betas"] = (0.8, 0.95)
            ref.param_groups[0]["betas"] = (0.8, 0.95)
        grad = torch.randn(5, 2, generator=g, dtype=torch.float64)
        a.grad, b.grad = grad.clone(), grad.clone()
        ours.step()
        ref.step()
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-12, atol=1e-14), it
    # old format: only the moments, powers implied by group["step"]
    import copy
    sd = copy.deepcopy(ours.state_dict())      # (state_dict() shares the per-parameter dicts with the optimiser)
    for st in sd["state"].values():
        for k in ("bias_pows", "betas", "betas_host"):
            st.pop(k, None)
    c = torch.nn.Parameter(a.detach().clone())
    loaded = RiemannianAdam([c], lr=0.05, eps=1e-7)
    loaded.load_state_dict(sd)
    grad = torch.randn(5, 2, generator=g, dtype=torch.float64)
    a.grad, c.grad = grad.clone(), grad.clone()
    ours.step()
    loaded.step()
    assert torch.allclose(a.detach(), c.detach(), rtol=1e-12, atol=1e-14)


def test_sort_batches_by_source():
    """data.sort_batches_by_source: inside every FULL batch the pairs are ordered by their first column (stable), every batch keeps
    its own triplets, the ragged tail is untouched."""
    import torch
    from sympa_amd import data
    g = torch.Generator().manual_seed(3)
    t = torch.stack((torch.randint(0, 50, (1030,), generator=g), torch.randint(0, 50, (1030,), generator=g),
                     torch.randint(1, 9, (1030,), generator=g)), 1)
    out = data.sort_batches_by_source(t, 256)
    assert out.shape == t.shape and torch.equal(out[1024:], t[1024:])
    for k in range(4):
        a, b = t[k * 256:(k + 1) * 256], out[k * 256:(k + 1) * 256]
        assert bool((b[1:, 0] >= b[:-1, 0]).all())
        assert torch.equal(b, a[torch.argsort(a[:, 0], stable=True)])
    assert torch.equal(data.sort_batches_by_source(t[:100], 256), t[:100])


def test_clock_stamps_are_paired_per_cu():
    """tools/clock_util.between (host side of C-ABI sympa_clock_stamp, bench.py `clock`): the shader-cycle counters of different CUs
    are not comparable, so only stamps taken on the SAME CU (XCC id + the CU / SH / SE bits of HW_ID) are paired; the clock is the
    median over the CUs seen by both stamps."""
    import os
    import sys
    from tests.helpers import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import clock_util

    def stamp(offsets, t_real, mhz):
        rows = torch.zeros(clock_util.BLOCKS, 3, dtype=torch.int64)
        for b in range(clock_util.BLOCKS):
            cu = b % 32                      # 32 CUs, 64 blocks each
            xcc = cu % 8
            hw = (cu // 8) << 8              # CU field of HW_ID (bits 8..11)
            rows[b, 0] = offsets[cu] + int(t_real * mhz[cu] / 100.0)
            rows[b, 1] = t_real + b          # blocks start a little apart (100 MHz ticks)
            rows[b, 2] = xcc | (hw << 8)
        return rows.reshape(-1)

    offsets = [10 ** 9 * (k + 1) for k in range(32)]         # wildly different counter origins per CU
    mhz = [2000.0 + 10.0 * k for k in range(32)]
    a = stamp(offsets, 1_000_000, mhz)
    b = stamp(offsets, 1_000_000 + 50_000, mhz)              # 500 us later
    # (the synthetic counter is linear in the real time of block 0: every CU's clock comes back exactly)
    r = clock_util.between(a, b)
    assert r["cus_paired"] == 32 and abs(r["region_us"] - 500.0) < 1e-6
    assert abs(r["mhz_min_cu"] - 2000.0) < 0.5 and abs(r["mhz_max_cu"] - 2310.0) < 0.5 and abs(r["mhz"] - 2160.0) < 0.5
    assert sorted(r["mhz_per_xcd"]) == [str(x) for x in range(8)]
    assert clock_util.between(None, b) is None
