#!/usr/bin/env python3
"""The secondary rows of bench.py's record ("plus the other configs as secondary rows"; round-5 review item 2): every
claim about a non-headline workload measured by the driver's own command, in the SAME JSON line as the headline.

    forward rows   configs[0..4] (BASELINE.json), each in two forms:
        "list"    Model.forward_batches(Model.prepare_batches(K batches))   -- the loop of Runner.evaluate (runner.py:124-135) as ONE call
        "single"  K calls of Model.forward under no_grad                     -- the call the reference makes (model.py:16-30), one per batch
    training rows  the whole training step of Runner.train_epoch (runner.py:98-118: forward, AverageDistortionLoss, backward,
                   clip, RiemannianSGD) of the headline, configs[3] and configs[4], replayed hipGraphs (sympa_amd/train_step.py)

Per row: ms_per_step (wall clock: synchronize, K steps, synchronize; median of 5), device_us_per_step (HIP events around back-to-back
repetitions: every kernel of a step, the pack's validity check included), launches_per_step, the contract fraction
(SURVEY 8d algorithmic bytes per pair x pairs / device time / 8 TB/s), parity.max_rel_err of the timed code's own output against the
oracle on a sample, packed_table.pack_us where a packed table is used.  `python tools/bench_rows.py` prints the rows alone."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12
FORWARD = (   # name, manifold, metric, dims, nodes, batch   (BASELINE.json configs[0..4])
    ("grid-upper-riem-n2-b512", "upper", "riem", 2, 125, 512),
    ("tree-upper-riem-n4-b8192", "upper", "riem", 4, 1093, 8192),
    ("margulis-bounded-finf-n4-b65536", "bounded", "finf", 4, 5041, 65536),
    ("cartesian-upper-riem-n8-b262144", "upper", "riem", 8, 45500, 262144),
    ("custom-spd-n16-b1048576", "spd", "riem", 16, 100000, 1048576),
)
TRAIN = (
    ("upper-riem-n4-b65536", "upper", "riem", 4, 5041, 65536),
    ("cartesian-upper-riem-n8-b262144", "upper", "riem", 8, 45500, 262144),
    ("custom-spd-n16-b1048576", "spd", "riem", 16, 100000, 1048576),
)


def bytes_per_pair(n, model):
    return 2 * 8 + 2 * ((1 if model == "spd" else 2) * n * n * 8) + 8


def _r(x, digits=5):
    """floats of the rows to five significant digits: the rows ride in bench.py's ONE JSON line"""
    return float(f"{x:.{digits}g}") if x is not None and x == x and abs(x) != float("inf") else x


LEGEND = {
    "forms": {"list": "Model.forward_batches(Model.prepare_batches(K batches)): the loop of Runner.evaluate (runner.py:124-135) as one call",
              "single": "Model.forward(batch) under no_grad, once per step: the call the reference makes (model.py:16-30)",
              "training_step": "sympa_amd.train_step.GraphedTrainStep: runner.py:98-118 as replayed hipGraphs (RiemannianSGD, max_grad_norm "
                               "50; headline: two kernels per step, deterministic accumulation; dims >= 7 / spd: one graph per step); "
                               "batches addressed by the device step counter where `windowed` is true"},
    "ms_per_step": "wall clock: synchronize, K steps, synchronize; median of 5",
    "device_us_per_step": "HIP events around back-to-back repetitions (>= 64 launches per group): every kernel of a step, the pack's "
                          "validity check included where a packed table is used; kernel_avg_us = the same per launch of the pair "
                          "kernel (a list of K <= 32 Siegel batches is ONE launch)",
    "roofline.frac": "SURVEY 8d algorithmic bytes per pair x pairs per step / device_us_per_step / 8 TB/s (training: two points read, two "
                     "gradient rows read-modify-written, ids + graph distance); frac_whole_job: the same with ms_per_step",
    "parity": "forward: the timed code's own output of batch 0 against oracle/siegel_oracle.py on a sample, relative, tol 1e-4.  training: "
              "the table gradient and the loss of the fused loss + backward kernels on 2 048 pairs of 400 rows against torch autograd of "
              "AverageDistortionLoss through the oracle (relative to the largest gradient entry); checker_noise_floor = the asymmetry "
              "of the oracle's own gradient; vs_svd_formulation = the same gradient against autograd through svdvals of "
              "L1^-1 (Z2 - Z1) L2^-T, a well-conditioned formulation that is not the reference's chain and adjudicates the few pairs per "
              "thousand on which the reference's 2n x 2n symeig backward is off by 1e-5 .. 5e-5",
    "packed_table": "pack_us = digest + unconditional pack (table changed); validity_check_us = digest + a pack kernel that returns at "
                    "once (table unchanged: part of every `single` step that uses the pack and of every `list` call)",
}


def _median(xs):
    xs = sorted(xs)
    return xs[(len(xs) - 1) // 2]


def _table(model, n, nodes, seed):
    from sympa_amd import data
    return data.spd_table(nodes, n, seed=seed) if model == "spd" else data.trained_like_table(nodes, n, model=model, seed=seed)


def _net(model, metric, n, nodes, table_cpu, dev, train_scale=False):
    import torch
    from sympa_amd.model import Model

    class A:
        manifold, dims, num_points = model, n, nodes
        scale_coef, scale_init = 1.0, 1.0
    A.metric, A.train_scale = metric, train_scale
    net = Model(A)
    with torch.no_grad():
        net.embeddings.embeds.data = table_cpu.clone()
    return net.to(dev)


def _time(run, dev, steps, launches_hint=1):
    """(wall ms per step: median of 5 repetitions of synchronize / K steps / synchronize; device us per step: HIP events around
    back-to-back repetitions, >= 64 launches per group where a step is short)."""
    import torch
    for _ in range(3):
        run()
    torch.cuda.synchronize(dev)
    wall = []
    for _ in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize(dev)
        wall.append(time.perf_counter() - t0)
    reps = max(1, -(-64 // max(1, launches_hint)))
    evs = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run()
        b.record()
        torch.cuda.synchronize(dev)
        evs.append(a.elapsed_time(b) / reps)
    return _median(wall) / steps * 1e3, _median(evs) / steps * 1e3


def _oracle_forward(model, metric, table_cpu, pairs_cpu):
    import torch
    from oracle import siegel_oracle as so
    with torch.no_grad():
        if model == "spd":
            return so.spd_model_forward(table_cpu, pairs_cpu, torch.ones(1, dtype=torch.float64), 1.0)
        return so.model_forward(table_cpu, pairs_cpu, model, metric)


def _parity(got, want):
    import torch
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    ok = bool(torch.isfinite(got).all())
    rel = float(((got - want).abs() / want.abs().clamp_min(1e-9)).max()) if ok else float("inf")
    return {"pairs": int(got.numel()), "max_rel_err": _r(rel), "ok": bool(ok and rel <= 1e-4)}


def forward_rows(name, model, metric, n, nodes, batch, dev, seed, steps, table_cpu=None):
    import torch
    from sympa_amd import data, ops
    table_cpu = _table(model, n, nodes, seed) if table_cpu is None else table_cpu
    net = _net(model, metric, n, nodes, table_cpu, dev)
    nb = 4
    batches = [data.sample_pairs(nodes, batch, j, seed).to(dev) for j in range(nb)]
    blist = [batches[i % nb] for i in range(steps)]
    olist = [torch.empty(batch, dtype=torch.float64, device=dev) for _ in range(nb)]
    plan = net.prepare_batches(blist, [olist[i % nb] for i in range(steps)])
    sample = min(batch, 256 if model == "spd" else 1024)
    want = _oracle_forward(model, metric, table_cpu, batches[0][:sample, :2].cpu())
    bpp = bytes_per_pair(n, model)
    pk = net.packed_table()
    rows = []
    last = {}

    def run_list():
        net.forward_batches(plan)

    def run_single():
        for i in range(steps):
            last[i % nb] = net(batches[i % nb])

    with torch.no_grad():
        for form, run, launches in (("list", run_list, -(-steps // ops.MAX_FUSED_BATCHES) if model != "spd" else steps),
                                    ("single", run_single, steps)):
            ms, dev_us = _time(run, dev, steps, launches)
            ops.check_status(dev)
            out0 = olist[0] if form == "list" else last[0]
            # (single calls of the upper model stay on the dense kernel while the pack's validity check is on: sympa_amd/model.py)
            packed_now = pk is not None and pk.key is not None and batch * (steps if form == "list" else 1) >= 4096 and \
                (form == "list" or model != "upper" or not pk.strict)
            row = {"workload": name, "kind": "forward", "form": form, "steps": steps, "pairs_per_step": batch,
                   "ms_per_step": _r(ms), "device_us_per_step": _r(dev_us), "kernel_avg_us": _r(dev_us * steps / launches),
                   "value": _r(batch / (ms * 1e-3)),
                   "roofline": {"bytes_per_pair": bpp, "frac": _r(bpp * batch / (dev_us * 1e-6) / HBM_PEAK),
                                "frac_whole_job": _r(bpp * batch / (ms * 1e-3) / HBM_PEAK)},
                   "packed": bool(packed_now), "parity": _parity(out0[:sample], want)}
            rows.append(row)
        if pk is not None and pk.key is not None:
            def repack():
                pk.invalidate()
                pk.ensure(net.embeddings.embeds)
            _, pack_us = _time(repack, dev, 1, 2)

            def check():
                pk.ensure(net.embeddings.embeds, strict=True)
            _, check_us = _time(check, dev, 1, 2)
            for r in rows:
                r["packed_table"] = {"pack_us": _r(pack_us), "validity_check_us": _r(check_us)}
    del net, plan
    return rows


def _oracle_grad(model, metric, table_cpu, trip_cpu, gd_cpu):
    """d (AverageDistortionLoss) / d table by torch autograd through the oracle (what the reference trains with)."""
    import torch
    from oracle import siegel_oracle as so
    tab = table_cpu.clone().requires_grad_(True)
    if model == "spd":
        d = so.spd_model_forward(tab, trip_cpu, torch.ones(1, dtype=torch.float64), 1.0)
    else:
        d = so.model_forward(tab, trip_cpu, model, metric)
    loss = so.distortion_loss(gd_cpu, d)
    loss.backward()
    g = tab.grad.detach()
    # the oracle's autograd runs through a 2n x 2n symeig (1 / eigenvalue-gap terms): its gradient of a symmetric argument comes out
    # asymmetric by its own rounding noise -- reported beside the comparison as the checker's noise floor
    noise = float((g - g.transpose(-1, -2)).abs().max() / g.abs().max().clamp_min(1e-300))
    return 0.5 * (g + g.transpose(-1, -2)), float(loss.detach()), noise


def _adjudicator_grad(metric, table_cpu, trip_cpu, gd_cpu):
    """NOT the reference's chain: the same loss through a well-conditioned formulation of the upper model's distance (singular values
    of E = L1^-1 (Z2 - Z1) L2^-T by torch.linalg.svdvals, v = 2 asinh(sigma / 2)) and torch autograd.  The reference's autograd runs
    through a 2n x 2n symeig whose backward divides by eigenvalue gaps: on a few pairs per thousand it is off by 1e-5 .. 5e-5
    (tests/test_backward.py adjudicates those with finite differences); this second checker tells such a pair from a real error."""
    import torch
    from oracle import siegel_oracle as so
    tab = table_cpu.clone().requires_grad_(True)
    z1, z2 = tab[trip_cpu[:, 0]], tab[trip_cpu[:, 1]]
    l1, l2 = torch.linalg.cholesky(z1[:, 1]), torch.linalg.cholesky(z2[:, 1])
    d = torch.complex(z2[:, 0] - z1[:, 0], z2[:, 1] - z1[:, 1])
    e = torch.linalg.solve_triangular(l1.to(d.dtype), d, upper=False)
    e = torch.linalg.solve_triangular(l2.to(d.dtype), e.transpose(-1, -2), upper=False).transpose(-1, -2)
    v = torch.sort(2.0 * torch.asinh(0.5 * torch.linalg.svdvals(e)), dim=-1)[0]
    dist = so.compute_metric(v, metric, None)
    so.distortion_loss(gd_cpu, dist).backward()
    g = tab.grad.detach()
    return 0.5 * (g + g.transpose(-1, -2))


def train_row(name, model, metric, n, nodes, batch, dev, seed, steps, table_cpu=None):
    import torch
    from sympa_amd import data, ops
    from sympa_amd.optim import RiemannianSGD
    from sympa_amd.train_step import GraphedTrainStep
    table_cpu = _table(model, n, nodes, seed) if table_cpu is None else table_cpu
    # ---- parity first, on its own small model: gradient of the fused loss + backward kernels on a sample against autograd
    # through the oracle (same table rows, 2 048 pairs)
    sample = 2048            # (>= 1 024 pairs: the split / three-kernel backward forms the large batches run)
    rows_used = min(nodes, 400)
    small = table_cpu[:rows_used].clone()
    g = torch.Generator().manual_seed(seed)
    trip = torch.stack((torch.randint(0, rows_used, (sample,), generator=g), torch.randint(0, rows_used, (sample,), generator=g),
                        torch.randint(1, 9, (sample,), generator=g)), 1)
    trip[:, 1] = torch.where(trip[:, 1] == trip[:, 0], (trip[:, 1] + 1) % rows_used, trip[:, 1])
    net = _net(model, metric, n, rows_used, small, dev, train_scale=False)
    net.embeddings.embeds.grad = None
    loss_dev = net.fused_loss_backward(trip.to(dev), trip[:, 2].to(torch.float64).to(dev))
    torch.cuda.synchronize(dev)
    ops.check_status(dev)
    want_g, want_loss, noise = _oracle_grad(model, metric, small, trip[:, :2], trip[:, 2].to(torch.float64))
    got_g = net.embeddings.embeds.grad.cpu()
    got_g = 0.5 * (got_g + got_g.transpose(-1, -2))
    rel = float((got_g - want_g).abs().max() / want_g.abs().max().clamp_min(1e-300))
    lrel = abs(float(loss_dev.cpu()) - want_loss) / max(abs(want_loss), 1e-300)
    adj = None
    if model == "upper":
        alt = _adjudicator_grad(metric, small, trip[:, :2], trip[:, 2].to(torch.float64))
        adj = float((got_g - alt).abs().max() / alt.abs().max().clamp_min(1e-300))
    parity = {"pairs": sample, "max_rel_err": _r(max(rel, lrel)), "grad_max_rel_err": _r(rel), "loss_rel_err": _r(lrel),
              "checker_noise_floor": _r(noise), "vs_svd_formulation": _r(adj),
              "ok": bool(max(rel, lrel) <= 1e-4)}
    del net
    # ---- the timed step
    net = _net(model, metric, n, nodes, table_cpu, dev, train_scale=True)
    opt = RiemannianSGD(net.parameters(), lr=1e-4)
    k_epoch = max(steps, 8)
    big = torch.stack((torch.randint(0, nodes, (batch * k_epoch,), generator=g), torch.randint(0, nodes, (batch * k_epoch,), generator=g),
                       torch.randint(1, 9, (batch * k_epoch,), generator=g)), 1).to(dev)
    two = model != "spd" and n <= 6
    step = GraphedTrainStep(net, opt, batch, 50.0, dev, two_kernels=two, deterministic=True if two else False, accumulate_loss=two)
    windowed = step.mode == "two_kernels" or step._classic_windowed()
    if windowed:

        def run():
            step.load_epoch(big[:batch * steps])
            step.run_steps(steps)
        # load_epoch (one sort + copies per EPOCH) stays outside the timed steps
        def timed():
            step.run_steps(steps)
        for _ in range(2):
            run()
        torch.cuda.synchronize(dev)
        wall, evs = [], []
        for _ in range(5):
            step.load_epoch(big[:batch * steps])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            timed()
            torch.cuda.synchronize(dev)
            wall.append(time.perf_counter() - t0)
        for _ in range(5):
            step.load_epoch(big[:batch * steps])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timed()
            b.record()
            torch.cuda.synchronize(dev)
            evs.append(a.elapsed_time(b))
        ms, dev_us = _median(wall) / steps * 1e3, _median(evs) / steps * 1e3
    else:
        if (model == "upper" and n == 8) or (model == "spd" and 9 <= n <= 16):
            big = data.sort_batches_by_source(big, batch)       # what the data pipeline does per epoch (sympa_amd/data.py)
        ids = [big[j * batch:(j + 1) * batch, :2].contiguous() for j in range(4)]
        gds = [big[j * batch:(j + 1) * batch, 2].to(torch.float64) for j in range(4)]

        def run():
            for i in range(steps):
                step(ids[i % 4], gds[i % 4])
        ms, dev_us = _time(run, dev, steps, steps * 4)
    ops.check_status(dev)
    finite = bool(torch.isfinite(net.embeddings.embeds.data).all())
    parity["table_finite_after_timed_steps"] = finite
    parity["ok"] = bool(parity["ok"] and finite)
    # contract: a training step reads both points and adds to both gradient rows (read-modify-write) per pair, plus ids,
    # graph distance and nothing else: 3 x the forward's point bytes + 32
    planes = 1 if model == "spd" else 2
    bpp = 3 * 8 + 2 * (planes * n * n * 8) + 2 * 2 * (planes * n * n * 8)
    return {"workload": name, "kind": "training_step", "form": "training_step", "windowed": bool(windowed), "steps": steps,
            "pairs_per_step": batch, "ms_per_step": _r(ms), "device_us_per_step": _r(dev_us), "value": _r(batch / (ms * 1e-3)),
            "roofline": {"bytes_per_pair": bpp, "frac": _r(bpp * batch / (dev_us * 1e-6) / HBM_PEAK)},
            "parity": parity}


def secondary_rows(dev, seed=42, steps=20, budget_s=150.0, log=None):
    """All rows, in a fixed order, until the time budget is spent (a row that did not fit is listed with "skipped")."""
    import torch
    t_start = time.perf_counter()
    rows = []
    tables = {}

    def tab(model, n, nodes):
        key = (model, n, nodes)
        if key not in tables:
            tables[key] = _table(model, n, nodes, seed)
        return tables[key]

    jobs = [("forward",) + w for w in FORWARD[:4]] + [("train",) + TRAIN[0], ("train",) + TRAIN[1],
                                                      ("forward",) + FORWARD[4], ("train",) + TRAIN[2]]
    for kind, name, model, metric, n, nodes, batch in jobs:
        left = budget_s - (time.perf_counter() - t_start)
        need = 45.0 if model == "spd" else 8.0
        if left < need:
            rows.append({"workload": name, "kind": "forward" if kind == "forward" else "training_step",
                         "skipped": f"time budget ({budget_s:.0f} s) spent"})
            continue
        t0 = time.perf_counter()
        try:
            if kind == "forward":
                new = forward_rows(name, model, metric, n, nodes, batch, dev, seed, steps, tab(model, n, nodes))
            else:
                new = [train_row(name, model, metric, n, nodes, batch, dev, seed, steps, tab(model, n, nodes))]
        except Exception as e:  # noqa: BLE001  (a secondary row must never take the headline record down)
            new = [{"workload": name, "kind": kind, "error": f"{type(e).__name__}: {e}"}]
        for r in new:
            r["seconds"] = round(time.perf_counter() - t0, 1)
        rows += new
        torch.cuda.empty_cache()
        if log:
            log(f"secondary {kind} {name}: {time.perf_counter() - t0:.1f} s")
    return rows


if __name__ == "__main__":
    import json
    import torch
    dev = torch.device("cuda:0")
    out = secondary_rows(dev, budget_s=float(os.environ.get("BUDGET_S", "400")), log=lambda s: print(s, file=sys.stderr, flush=True))
    for r in out:
        if "skipped" in r or "error" in r:
            print(json.dumps(r))
            continue
        pt = r.get("packed_table")
        print(f"{r['workload']:34s} {r['kind']:13s} {r['form'][:6]:6s}  {r['ms_per_step'] * 1e3:9.1f} us/step wall  {r['device_us_per_step']:9.1f} us device  "
              f"frac {r['roofline']['frac']:.3f}  parity {r['parity']['max_rel_err']:.1e}" +
              (f"  pack {pt['pack_us']:.1f} us check {pt['validity_check_us']:.1f} us" if pt else ""), flush=True)
    print(json.dumps(out))
