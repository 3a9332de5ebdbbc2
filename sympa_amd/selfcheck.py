"""Numerical gate of the sixteen- / eight-lanes-per-pair kernels (inline-asm DPP, csrc/spd_coop.hpp).

The compiler does not see the cross-lane reads inside those asm statements; the build scans every translation unit's
ISA for the DPP hazard (tools/check_dpp_hazards.py), but docs/DESIGN_rounds_3_4.md section 13 records kernels that scanned clean and
were still wrong.  So the scan is not trusted alone: the first time an instantiation (family, model, n) of that layout
is used on a device, a small fixed batch goes through it AND through the one-lane-per-pair kernel of the same
arithmetic; on disagreement the instantiation is routed to the one-lane kernel for the rest of the process
(C-ABI sympa_set_instance_fallback), a warning is issued and the case is listed in `FAILURES` (tools/gpu_check.sh fails
on a non-empty list).  Nothing here is a CPU path: both sides of the comparison are HIP kernels of this library.

SYMPA_SELFCHECK=0 switches the gate off.  Inside a stream capture the check is postponed (it synchronises)."""
import os
import warnings

import torch

from sympa_amd import _lib

SIEGEL_FWD, SIEGEL_BWD, SIEGEL_TABLE, SPD_FWD, SPD_BWD, SPD_TABLE = range(6)
NAMES = ("siegel forward", "siegel backward", "siegel table ops", "spd forward", "spd backward", "spd table ops")
# dims whose DEFAULT dispatch reaches the layout (csrc dispatchers; include/sympa_hip.h SYMPA_FAMILY_*)
RANGE = {SIEGEL_FWD: (9, 16), SIEGEL_BWD: (7, 16), SIEGEL_TABLE: (7, 16), SPD_FWD: (6, 16), SPD_BWD: (3, 16),
         SPD_TABLE: (3, 16)}
MODEL_IDS = {"upper": 0, "bounded": 1, "spd": 0}

ENABLED = os.environ.get("SYMPA_SELFCHECK", "1") != "0"
FAILURES = []          # (family name, model, n, device index, detail)
CHECKED = set()        # (family, model, n, device index)


def _close(a, b, rel):
    """max |a - b| <= rel * max |b| (+ tiny), NaN-safe: a NaN on either side is a disagreement."""
    if a.shape != b.shape:
        return False, "shape"
    if not (torch.isfinite(a).all() and torch.isfinite(b).all()):
        return False, "non-finite"
    err = float((a - b).abs().max()) if a.numel() else 0.0
    ref = float(b.abs().max()) if b.numel() else 0.0
    return err <= rel * ref + 1e-300, f"max abs diff {err:.3e} against max {ref:.3e}"


class _OneLane:
    """Context: route (family, model, n) to the one-lane kernels, as a failed check would."""

    def __init__(self, family, model, n):
        self.args = (family, MODEL_IDS[model], n)

    def __enter__(self):
        _lib.check(_lib.load().sympa_set_instance_fallback(*self.args, 1))

    def __exit__(self, *exc):
        _lib.check(_lib.load().sympa_set_instance_fallback(*self.args, 0))


def _inputs(model, n, dev, b):
    from sympa_amd import data
    rows = 48
    table = (data.spd_table(rows, n, scale=0.3, seed=77) if model == "spd"
             else data.trained_like_table(rows, n, model=model, seed=77)).to(dev)
    trip = data.sample_pairs(rows, b, 0, 77)
    gd = (1.0 + (trip[:, 0] * 7 + trip[:, 1]) % 5).to(torch.float64)
    return table, trip.to(dev), gd.to(dev)


def _check_siegel_fwd(model, n, dev):
    from sympa_amd import ops
    table, trip, _ = _inputs(model, n, dev, 333)
    w = torch.linspace(-0.3, 1.2, n, device=dev)
    res = []
    for metric in ("riem", "wsum"):
        fast = ops.model_forward(table, trip, model, metric, w)
        with _OneLane(SIEGEL_FWD, model, n):
            slow = ops.model_forward(table, trip, model, metric, w)
        res.append(_close(fast, slow, 1e-8))
    return res


def _check_siegel_bwd(model, n, dev):
    from sympa_amd import ops
    res = []
    for b in (333,):
        table, trip, gd = _inputs(model, n, dev, b)
        sc = torch.ones(1, dtype=torch.float64, device=dev)

        def run():
            rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=dev)
            l1 = torch.zeros(1, dtype=torch.float64, device=dev)
            ops.model_loss_backward_rows(table, trip, gd, rows, l1, model, "riem", scale=sc)
            dense = torch.zeros_like(table)
            l2 = torch.zeros(1, dtype=torch.float64, device=dev)
            ops.model_loss_backward(table, trip, gd, dense, l2, model, "riem", scale=sc)
            return rows, dense, torch.cat((l1, l2))

        fast = run()
        with _OneLane(SIEGEL_BWD, model, n):
            slow = run()
        res += [_close(f, s, 1e-6) for f, s in zip(fast, slow)]
    return res


def _check_siegel_table(model, n, dev):
    from sympa_amd import ops
    table, _, _ = _inputs(model, n, dev, 8)
    g = torch.Generator().manual_seed(5)
    grad = torch.randn(table.shape, generator=g, dtype=torch.float64)
    grad = (0.5 * (grad + grad.transpose(-1, -2))).to(dev)

    def run():
        out = [ops.egrad2rgrad(table, grad, model), ops.tangent_sqnorm(table, grad, model), ops.projx(table, model)]
        for lr in (1e-3, 0.5):      # the second one pushes rows out of the eps-interior: the gated exact projection runs
            t = table.clone()
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            ops.rsgd_step_(t, grad, model, lr, counter=cnt)
            out += [t, cnt.to(torch.float64)]
        return out

    fast = run()
    with _OneLane(SIEGEL_TABLE, model, n):
        slow = run()
    return [_close(f, s, 1e-7) for f, s in zip(fast, slow)]


def _check_spd_fwd(model, n, dev):
    from sympa_amd import ops
    table, trip, _ = _inputs("spd", n, dev, 333)
    fast = ops.spd_model_forward(table, trip)
    # the PACKED instantiation (spd16_coop_kernel<M, true>) is a separate inline-asm DPP binary (reduce_pair_front_factored): checked
    # through the raw C entries, whatever ops.SPD_PACKED_DIMS offers
    lib = _lib.load()
    need = int(lib.sympa_spd_table_pack_bytes(table.shape[0], n))
    packed = None
    if need > 0:
        pack = torch.empty(need, dtype=torch.uint8, device=dev)
        packed = torch.empty(trip.shape[0], dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.sympa_spd_table_pack(table.data_ptr(), table.shape[0], n, pack.data_ptr(), need, None, stream))
        tp, stride = trip.data_ptr(), trip.stride(0)
        _lib.check(lib.sympa_spd_model_forward_packed(pack.data_ptr(), need, table.shape[0], n, tp, stride, tp + 8, stride,
                                                      trip.shape[0], None, 1.0, packed.data_ptr(), ops._status_buf(dev).data_ptr(),
                                                      0, stream))
    with _OneLane(SPD_FWD, "spd", n):
        slow = ops.spd_model_forward(table, trip)
    return [_close(fast, slow, 1e-8)] + ([] if packed is None else [_close(packed, slow, 1e-8)])


def _check_spd_bwd(model, n, dev):
    from sympa_amd import ops
    res = []
    # 333: the single-round kernel (also what finishes flagged chunks behind the three-kernel path); 8 197: n = 16 the
    # three-kernel path (eigenvectors one pair per lane, from 1 024 pairs on), other n the QL of two rounds run together
    for b in (333, 8192 + 5):
        table, trip, gd = _inputs("spd", n, dev, b)
        idx = torch.cat((trip[:, 0], trip[:, 1])).contiguous()

        def run(one_lane):
            l1 = torch.zeros(1, dtype=torch.float64, device=dev)
            rows, _ = ops.spd_backward_rows(table, table, trip, graph_dist=gd, loss=l1)
            dense = torch.zeros_like(table)
            l2 = torch.zeros(1, dtype=torch.float64, device=dev)
            if one_lane:            # no one-lane kernel with the scatter inside: rows + scatter is its counterpart
                ops.scatter_add_flat_rows_(dense, rows[:2 * b].reshape(2 * b, -1), idx)
                l2 = l1.clone()
            else:
                ops.spd_loss_backward(table, trip, dense, graph_dist=gd, loss=l2)
            return rows[:2 * b], dense, torch.cat((l1, l2))

        fast = run(False)
        with _OneLane(SPD_BWD, "spd", n):
            slow = run(True)
        res += [_close(f, s, 1e-6) for f, s in zip(fast, slow)]
    return res


def _check_spd_table(model, n, dev):
    from sympa_amd import ops
    table, _, _ = _inputs("spd", n, dev, 8)
    g = torch.Generator().manual_seed(5)
    grad = torch.randn(table.shape, generator=g, dtype=torch.float64)
    grad = (0.5 * (grad + grad.transpose(-1, -2))).to(dev)

    def run():
        t = table.clone()
        ops.spd_rsgd_step_(t, grad, 1e-2)
        return [ops.spd_egrad2rgrad(table, grad), t]

    fast = run()
    with _OneLane(SPD_TABLE, "spd", n):
        slow = run()
    return [_close(f, s, 1e-7) for f, s in zip(fast, slow)]


_CHECKS = {SIEGEL_FWD: _check_siegel_fwd, SIEGEL_BWD: _check_siegel_bwd, SIEGEL_TABLE: _check_siegel_table,
           SPD_FWD: _check_spd_fwd, SPD_BWD: _check_spd_bwd, SPD_TABLE: _check_spd_table}


def ensure(family, model, n, dev):
    """Runs the check of (family, model, n) once per device; returns True when the fast kernels are in use."""
    lo, hi = RANGE[family]
    if not ENABLED or n < lo or n > hi:
        return True
    key = (family, model, n, dev.index if dev.index is not None else torch.cuda.current_device())
    if key in CHECKED:
        return True
    if torch.cuda.is_current_stream_capturing():
        return True                      # postponed: the check synchronises
    CHECKED.add(key)                     # first: the check itself goes through the ops wrappers
    lib = _lib.load()
    if lib.sympa_get_instance_fallback(family, MODEL_IDS[model], n):
        return False                     # already routed to the one-lane kernels (set by the caller / another device)
    from sympa_amd import ops
    status = ops._status_buf(dev)
    with torch.cuda.device(dev), torch.no_grad():
        pending = status.clone()         # status bits the caller has not read yet are kept
        status.zero_()
        results = _CHECKS[family](model, n, dev)
        torch.cuda.synchronize(dev)
        status_ok = int(status[0]) == 0  # the fixed inputs are valid points: a raised status is a failure too
        status.copy_(pending)
    bad = [d for ok, d in results if not ok]
    if bad or not status_ok:
        _lib.check(lib.sympa_set_instance_fallback(family, MODEL_IDS[model], n, 1))
        detail = "; ".join(bad) if bad else "status word raised"
        FAILURES.append((NAMES[family], model, n, dev.index, detail))
        warnings.warn(f"sympa_amd: the {NAMES[family]} kernel of the lanes-per-pair layout for model={model} dims={n} disagrees "
                      f"with the one-lane kernel on this build ({detail}); using the one-lane kernel instead", RuntimeWarning)
        return False
    return True


def check_all(dev=None, families=None, dims=None):
    """Every instantiation of every family (tools/gpu_check.sh, tests): returns the list of failures."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
    for fam in (range(6) if families is None else families):
        lo, hi = RANGE[fam]
        models = ("spd",) if fam >= SPD_FWD else ("upper", "bounded")
        for model in models:
            for n in range(lo, hi + 1):
                if dims is None or n in dims:
                    ensure(fam, model, n, dev)
    return list(FAILURES)
