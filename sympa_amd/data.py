"""Host-side input generation for the hot path: integer/index work, bit-exact.

  * counter-based RNG (splitmix64 keyed by (seed, stream, counter)) so that the CPU run and every
    rank of a 1/2/4/8-GPU run see identical bytes, independent of torch RNG streams;
  * synthetic embedding tables: the reference init distribution (upper_half.py:116-131) and a
    'trained-like' table whose distances span O(0.1 .. 10);
  * pair batches: sampled (i, j != i) pairs, or all (i < j, graph distance) triplets of a graph in
    lexicographic order with BFS distances (what preprocess.py:101-126 produces through networkit);
  * DistributedSampler semantics for sharding a triplet list over ranks (train.py:105-110).
"""
import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return x ^ (x >> np.uint64(31))


def keyed_u64(seed, stream, counters):
    """u64 = splitmix64(splitmix64(seed*2^32 + stream) ^ counter): one independent value per counter."""
    c = np.asarray(counters, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64((int(seed) << 32 | int(stream)) & 0xFFFFFFFFFFFFFFFF) + np.zeros(1, np.uint64))[0]
    return splitmix64(c ^ key)


def keyed_uniform(seed, stream, counters):
    """Uniform doubles in [0,1) with 53 random bits."""
    return (keyed_u64(seed, stream, counters) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def keyed_normal(seed, stream, counters):
    """Standard normals by Box-Muller on two keyed uniforms (streams 2s, 2s+1)."""
    u1 = keyed_uniform(seed, 2 * stream, counters)
    u2 = keyed_uniform(seed, 2 * stream + 1, counters)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)


def _sym(a):
    return 0.5 * (a + np.swapaxes(a, -1, -2))


def init_table(num_points, dims, seed=42, eps=1e-3):
    """Reference init distribution: X = sym(U(-eps,eps)), Y = I + sym(U(-eps,eps)) -> [N,2,n,n] fp64."""
    cnt = np.arange(num_points * dims * dims, dtype=np.uint64)
    x = (keyed_uniform(seed, 1, cnt) * 2 - 1) * eps
    y = (keyed_uniform(seed, 2, cnt) * 2 - 1) * eps
    x = _sym(x.reshape(num_points, dims, dims))
    y = np.eye(dims)[None] + _sym(y.reshape(num_points, dims, dims))
    return torch.from_numpy(np.stack((x, y), 1))


def trained_like_table(num_points, dims, scale=0.3, seed=42, model="upper"):
    """X = sym(N*s), Y = expm(sym(N*s)): points spread over the manifold (distances O(0.1..10)).
    model='bounded' returns the Cayley image (exactly symmetric)."""
    cnt = np.arange(num_points * dims * dims, dtype=np.uint64)
    x = _sym((keyed_normal(seed, 3, cnt) * scale).reshape(num_points, dims, dims))
    a = _sym((keyed_normal(seed, 4, cnt) * scale).reshape(num_points, dims, dims))
    lam, vec = np.linalg.eigh(a)
    y = _sym((vec * np.exp(lam)[:, None, :]) @ np.swapaxes(vec, -1, -2))
    if model == "upper":
        return torch.from_numpy(np.stack((x, y), 1))
    z = x + 1j * y
    eye = np.eye(dims)[None]
    w = (z - 1j * eye) @ np.linalg.inv(z + 1j * eye)      # cayley_transform.py:10-24
    w = 0.5 * (w + np.swapaxes(w, -1, -2))
    return torch.from_numpy(np.stack((w.real, w.imag), 1))


def spd_table(num_points, dims, scale=0.1, seed=42):
    """[N, n, n] SPD points expm(sym(N*s)) for the `spd` model (configs[4]); same keyed RNG as the Siegel tables."""
    cnt = np.arange(num_points * dims * dims, dtype=np.uint64)
    a = _sym((keyed_normal(seed, 5, cnt) * scale).reshape(num_points, dims, dims))
    lam, vec = np.linalg.eigh(a)
    return torch.from_numpy(_sym((vec * np.exp(lam)[:, None, :]) @ np.swapaxes(vec, -1, -2)))


def sample_pairs(num_points, batch, batch_id=0, seed=42):
    """int64 [batch,2]: pair k of batch `batch_id` is (i, (i + 1 + U[0, N-2]) mod N), i != j."""
    base = np.uint64(batch_id) * np.uint64(batch)
    cnt = base + np.arange(batch, dtype=np.uint64)
    i = (keyed_u64(seed, 10, cnt) % np.uint64(num_points)).astype(np.int64)
    off = (keyed_u64(seed, 11, cnt) % np.uint64(num_points - 1)).astype(np.int64)
    j = (i + 1 + off) % num_points
    return torch.from_numpy(np.stack((i, j), 1))


def graph_triplets(graph):
    """All (i, j, d) with i < j, 0 < d < inf of a networkx graph whose nodes are relabelled by
    sorted() (preprocess.py:152-153,101-126); lexicographic order; BFS hop distances (exact ints).
    Returns (int64 [T,3] tensor, id2node dict)."""
    import networkx as nx
    from scipy.sparse.csgraph import shortest_path

    nodes = sorted(graph.nodes())
    id2node = {i: node for i, node in enumerate(nodes)}
    g = nx.convert_node_labels_to_integers(graph, ordering="sorted")
    adj = nx.to_scipy_sparse_array(nx.Graph(g), nodelist=range(len(nodes)), weight=None, format="csr")
    adj.setdiag(0)
    adj.eliminate_zeros()
    dist = shortest_path(adj, method="D", unweighted=True, directed=False)
    iu, ju = np.triu_indices(len(nodes), k=1)
    d = dist[iu, ju]
    keep = np.isfinite(d) & (d > 0)
    trip = np.stack((iu[keep], ju[keep], d[keep].astype(np.int64)), 1).astype(np.int64)
    return torch.from_numpy(trip), id2node


def named_graph(name):
    """Graphs of BASELINE.json's configs (preprocess.py:12-73)."""
    import networkx as nx
    if name == "grid3d-125":
        return nx.grid_graph(dim=[5, 5, 5])
    if name == "tree-b3-h6":
        return nx.balanced_tree(3, 6)
    if name == "margulis-71":
        return nx.margulis_gabber_galil_graph(71)
    raise KeyError(name)


def distributed_sampler_indices(dataset_len, num_replicas, rank, epoch=0, seed=0, shuffle=True, drop_last=False):
    """Index list torch.utils.data.DistributedSampler yields (the sharding train.py:105-110 uses):
    randperm(seed+epoch) -> pad by wrapping to a multiple of world -> indices[rank::world]."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        indices = torch.randperm(dataset_len, generator=g).tolist()
    else:
        indices = list(range(dataset_len))
    if drop_last and dataset_len % num_replicas != 0:
        num_samples = -(-(dataset_len - num_replicas) // num_replicas)
    else:
        num_samples = -(-dataset_len // num_replicas)
    total = num_samples * num_replicas
    if not drop_last:
        pad = total - len(indices)
        if pad <= len(indices):
            indices += indices[:pad]
        else:
            indices += (indices * (-(-pad // len(indices))))[:pad]
    else:
        indices = indices[:total]
    return indices[rank:total:num_replicas]


# ---------------------------------------------------------------------------------------------------
# On-disk formats either side of the path
# ---------------------------------------------------------------------------------------------------
def save_preprocessed(path, triplets, id2node):
    """Writes the reference's `preprocessed-data.pt` (preprocess.py:165-171): a torch-pickled dict with
    "triplets" = a Python set of (src, dst, distance) tuples and "id2node" = {id: node label}."""
    trip = triplets.tolist() if torch.is_tensor(triplets) else list(triplets)
    as_set = {(int(i), int(j), (int(d) if float(d).is_integer() else float(d))) for i, j, d in trip}
    torch.save({"triplets": as_set, "id2node": dict(id2node)}, path)


def load_preprocessed(path):
    """Reads `preprocessed-data.pt` the way train.py:80-97 does, but into canonical (lexicographic) order
    instead of CPython set-iteration order: returns (src_dst_ids int64 [T,2], distances fp64 [T], id2node)."""
    blob = torch.load(path, weights_only=False)
    trip = sorted(blob["triplets"])
    ids = torch.tensor([(s, d) for s, d, _ in trip], dtype=torch.int64).reshape(-1, 2)
    dist = torch.tensor([float(w) for _, _, w in trip], dtype=torch.float64)
    return ids, dist, blob["id2node"]


def scale_triplet_distances(distances):
    """utils.scale_triplets (sympa/utils.py:71-82): squared distances divided by their maximum."""
    sq = distances.to(torch.float64) ** 2
    return sq / sq.max()


def save_checkpoint(path, model, id2node):
    """runner.py:156-160: {"model": ddp_model.state_dict(), "id2node": ...}; DDP prefixes keys with "module."."""
    state = {f"module.{k}": v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"model": state, "id2node": dict(id2node)}, path)


def load_checkpoint(path, model):
    """train.py:60-62 (`--load_model`): accepts state dicts with or without DDP's "module." prefix."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in blob["model"].items()}
    model.load_state_dict(state)
    return blob.get("id2node")


def sort_batches_by_source(triplets, batch_size):
    """The epoch's triplets [T, >=2] with the pairs INSIDE every full batch ordered by their first column (stable; the ragged tail
    is left alone).  A training step sums the gradients of its batch, so the order inside a batch is free -- and the scatter of
    the split backward (dims 7, 8; csrc/siegel_bwd_split_kernel.hpp) adds consecutive pairs with the same source row as one
    atomic instruction instead of one per pair: fused step at configs[3] 0.80 -> 0.69 ms per 262 144 pairs.  One sort per epoch
    (train.py:105-110 reshuffles the sampler every epoch; this runs behind it)."""
    total = triplets.shape[0]
    full = (total // int(batch_size)) * int(batch_size)
    if full == 0:
        return triplets
    src = triplets[:full, 0]
    span = int(src.max().item()) + 1 if src.numel() else 1
    key = (torch.arange(full, device=triplets.device) // int(batch_size)) * span + src
    order = torch.argsort(key, stable=True)
    out = triplets.clone()
    out[:full] = triplets[:full].index_select(0, order)
    return out
