#!/usr/bin/env python3
"""What the device-side validity of a packed table costs (round 6: C-ABI sympa_table_digest / sympa_table_pack_refresh).
    python tools/pack_refresh_time.py
Per shape (configs[3]: upper n = 8, 45 500 rows, 262 144 pairs; configs[4]: spd n = 16, 100 000 rows, 1 048 576 pairs; a bounded
n = 7 shape): the digest kernel alone, refresh over an UNCHANGED table (digest + a pack kernel whose blocks return at once),
refresh over a CHANGED table (digest + pack), the unconditional pack, the packed forward alone and with the strict check in front
(what one Model.forward under no_grad costs).  HIP events around groups of sequential launches, median of 12 groups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sympa_amd import _lib, data, ops  # noqa: E402

dev = torch.device("cuda:0")
G = 8


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(G):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / G)
    ts.sort()
    return ts[len(ts) // 2]


lib = _lib.load()
for model, n, nodes, pairs in (("upper", 8, 45500, 262144), ("bounded", 7, 45500, 262144), ("spd", 16, 100000, 1048576)):
    spd = model == "spd"
    table = (data.spd_table(nodes, n, scale=0.3, seed=42) if spd else data.trained_like_table(nodes, n, model=model, seed=42)).to(dev)
    trip = data.sample_pairs(nodes, pairs, 0, 42).to(dev)
    out = torch.empty(pairs, dtype=torch.float64, device=dev)
    pk = (ops.SpdPackedTable() if spd else ops.PackedTable(model)).ensure(table)
    state = torch.zeros(4096, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nbytes = table.numel() * 8

    def digest():
        lib.sympa_table_digest(table.data_ptr(), nbytes, state.data_ptr(), 0, stream)

    def refresh_same():
        pk.ensure(table, strict=True)

    def refresh_changed():
        table.view(-1)[7] += 1e-13           # (version counter moves too: forced; the digest runs all the same)
        pk.ensure(table, strict=True)

    def pack_only():
        if spd:
            lib.sympa_spd_table_pack(table.data_ptr(), nodes, n, pk.pack.data_ptr(), pk.bytes, None, stream)
        else:
            lib.sympa_table_pack(table.data_ptr(), nodes, n, ops.MODEL_IDS[model], pk.pack.data_ptr(), pk.bytes, None, stream)

    def fwd():
        if spd:
            ops.spd_model_forward_packed(pk, trip, out=out)
        else:
            ops.model_forward_packed(pk, trip, "riem", out=out)

    def strict_fwd():
        pk.ensure(table, strict=True)
        fwd()

    t = {k: timed(f) for k, f in (("digest", digest), ("refresh_unchanged", refresh_same), ("refresh_changed", refresh_changed),
                                  ("pack", pack_only), ("forward", fwd), ("strict_forward", strict_fwd))}
    ops.check_status(dev)
    print(f"{model:8s} n={n:2d} rows={nodes:6d} table {nbytes / 1e6:6.1f} MB  pairs={pairs:7d}:  digest {t['digest']:6.1f} us "
          f"({nbytes / t['digest'] / 1e6:5.2f} TB/s)   refresh unchanged {t['refresh_unchanged']:6.1f}   refresh changed "
          f"{t['refresh_changed']:6.1f}   pack alone {t['pack']:6.1f}   packed forward {t['forward']:7.1f}   with the strict check "
          f"{t['strict_forward']:7.1f} us (+{100 * (t['strict_forward'] / t['forward'] - 1):.1f} %)", flush=True)
