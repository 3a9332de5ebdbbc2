"""Torch-facing entry points of the hot path.  PyTorch only supplies device memory and the
current HIP stream; all arithmetic happens in libsympa_hip.so through the C-ABI
(include/sympa_hip.h).  There is no CPU path: CPU tensors raise.

Error contract: the reference raises AssertionError from inside dist()
(siegel_manifold.py:64-66), which costs host syncs per call.  Here range violations are
accumulated in a per-device status word; `check_status()` (or `set_debug(True)`) surfaces them as
the same AssertionError / IndexError lazily."""
import ctypes
import os

import torch

from sympa_amd import _lib
from sympa_amd import selfcheck as _sc
from sympa_amd.config import EPS

MODEL_IDS = {"upper": 0, "bounded": 1}
METRIC_IDS = {"riem": 0, "fone": 1, "finf": 2, "fmin": 3, "wsum": 4}

ST_NOT_PD, ST_NONFINITE, ST_BAD_INDEX, ST_NO_CONVERGENCE = 1, 2, 4, 8

_status = {}   # device index -> int32[2] tensor
_debug = False


def set_debug(flag: bool):
    """When on, every call synchronises and raises like the reference's in-line asserts."""
    global _debug
    _debug = bool(flag)


def _status_buf(device):
    buf = _status.get(device)
    if buf is None:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        buf = _status.get(idx)
        if buf is None:
            buf = torch.zeros(2, dtype=torch.int32, device=device)
            _status[idx] = buf
        _status[device] = buf
    return buf


def check_status(device=None, reset=True):
    """Reads the device status word (one host sync) and raises what the reference would have."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    buf = _status_buf(torch.device(device))
    bits, count = (int(x) for x in buf.tolist())
    if reset and bits:
        buf.zero_()
    if bits & ST_BAD_INDEX:
        raise IndexError(f"index out of range in Model.forward gather ({count} pairs flagged)")
    if bits & (ST_NOT_PD | ST_NONFINITE | ST_NO_CONVERGENCE):
        raise AssertionError(
            f"Siegel distance: {count} pairs outside the manifold / non-finite (status bits {bits}); "
            "reference: 'assert 0 <= eigvalues <= 1' (siegel_manifold.py:64-66)")
    return bits, count


def _gate(family, model, n, dev):
    """First use of a lanes-per-pair kernel instantiation on a device: compare it with the one-lane kernel
    (sympa_amd/selfcheck.py).  One tuple lookup afterwards; dims below the family's range cost one comparison."""
    if n >= _sc.RANGE[family][0] and (family, model, n, dev.index) not in _sc.CHECKED:
        _sc.ensure(family, model, n, dev)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise _lib.SympaHipError(f"{name} is on {t.device}: the Siegel-distance path runs on the GPU only "
                                 "(HIP kernels, no CPU fallback)")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _weights(metric, weights, n, device):
    if metric != "wsum":
        return None
    if weights is None:
        raise ValueError("metric 'wsum' needs its weights")
    w = weights.detach().reshape(-1).to(device=device, dtype=torch.float64).contiguous()
    if w.numel() != n:
        raise ValueError(f"wsum weights must have {n} entries")
    return w


FLAG_LOW_LDS = 1
FLAG_GENERIC = 2      # spd: force the runtime-n one-lane-per-pair kernel
FLAG_ANY_ORDER = 4    # forward: dispatch without the in-order barrier bit (independent batches of one stream overlap)
FLAG_COOP = 32        # dims 6, 8: force the sixteen-lanes-per-pair kernel (A/B)
FLAG_NO_SYMMETRY = 16  # all_pairs_dist(packed=True): evaluate (i, j) and (j, i) separately
FLAG_SPLIT = 64        # backward, dims 5..8: the split (two-kernel) backward wherever it is built (default: upper model, dims 7, 8)
FLAG_FUSE = 8         # BatchedForward: up to MAX_FUSED_BATCHES consecutive batches per kernel launch
FLAG_MERGE_SRC = 128  # model_train_backward(grad_rows=...), dims <= 6: runs of equal source ids inside a wave leave as ONE row
MAX_FUSED_BATCHES = 32


def siegel_dist_forward(z1, z2, model="upper", metric="riem", weights=None, eps=None, return_vvd=False, flags=0):
    """manifold.dist for pre-gathered points z1, z2 [b,2,n,n] fp64 on the GPU -> [b]
    (C-ABI sympa_siegel_dist_fwd; reference siegel_manifold.py:41-72 / bounded_domain.py:27-39)."""
    lib = _lib.load()
    _need_gpu(z1, "z1"); _need_gpu(z2, "z2")
    if z1.dtype != torch.float64 or z2.dtype != torch.float64:
        raise TypeError("points must be float64 (reference default dtype, config.py:17-18)")
    if z1.shape != z2.shape or z1.dim() != 4 or z1.shape[1] != 2 or z1.shape[2] != z1.shape[3]:
        raise ValueError(f"expected two [b,2,n,n] tensors, got {tuple(z1.shape)} and {tuple(z2.shape)}")
    z1 = z1.detach().contiguous()
    z2 = z2.detach().contiguous()
    b, _, n, _ = z1.shape
    out = torch.empty(b, dtype=torch.float64, device=z1.device)
    vvd = torch.empty(b, n, dtype=torch.float64, device=z1.device) if return_vvd else None
    _gate(_sc.SIEGEL_FWD, model, n, z1.device)
    w = _weights(metric, weights, n, z1.device)
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(z1.device)
    with torch.cuda.device(z1.device):
        rc = lib.sympa_siegel_dist_fwd(_ptr(z1), _ptr(z2), b, n, MODEL_IDS[model], METRIC_IDS[metric],
                                       _ptr(w), eps, _ptr(out), _ptr(vvd), _ptr(st), int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(z1.device)
    return (out, vvd) if return_vvd else out


def model_forward(table, triplets, model="upper", metric="riem", weights=None, scale=None, scale_coef=1.0,
                  eps=None, out=None, flags=0):
    """Fused Model.forward (C-ABI sympa_model_forward; reference model.py:16-41): gathers table rows
    triplets[:,0] / triplets[:,1] inside the kernel, returns dist * clamp_min(scale/scale_coef, 0.1).
    The argument checks are kept cheap: this is called once per batch and the kernel takes ~7 us.  Where the thin torch
    binding is built (sympa_amd/_fast, csrc/torch_binding.cpp) the checks, the output allocation and the stream lookup
    happen in C++ and the call costs a third of the ctypes route (profiles/r04_host_call.txt); both call the same C-ABI entry
    of the same library."""
    fast = _lib.fast()
    if fast is not None and table.is_cuda and triplets.is_cuda and table.dim() == 4:
        n = table.shape[2]
        dev = table.device
        if n > 8:
            _gate(_sc.SIEGEL_FWD, model, n, dev)
        if not table.is_contiguous():
            table = table.contiguous()
        if triplets.dim() == 2 and triplets.stride(1) != 1:
            triplets = triplets.contiguous()
        w = _weights(metric, weights, n, dev) if metric == "wsum" else None
        if scale is not None and (not scale.is_cuda or scale.dtype is not torch.float64 or scale.device != dev):
            scale = scale.detach().to(device=dev, dtype=torch.float64)
        try:
            out = fast.model_forward(table, triplets, MODEL_IDS[model], METRIC_IDS[metric], w, scale, float(scale_coef),
                                     1e-5 if eps is None else float(eps), _status_buf(dev), int(flags), out)
        except RuntimeError as e:            # one error type for both bindings
            if type(e) is RuntimeError:
                raise _lib.SympaHipError(str(e).split("\n")[0]) from None
            raise
        if _debug:
            check_status(dev)
        return out
    lib = _lib.load()
    if not (table.is_cuda and triplets.is_cuda):
        _need_gpu(table, "table"); _need_gpu(triplets, "triplets")
    if table.dtype != torch.float64:
        raise TypeError("table must be float64")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
        raise TypeError("triplets must be an int64 [b, >=2] tensor (src, dst[, graph_distance])")
    shape = table.shape
    if len(shape) != 4 or shape[1] != 2 or shape[2] != shape[3]:
        raise ValueError(f"table must be [N,2,n,n], got {tuple(shape)}")
    tab = table if table.is_contiguous() else table.contiguous()
    num_rows, n = shape[0], shape[2]
    b = triplets.shape[0]
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    dev = tab.device
    if out is None:
        out = torch.empty(b, dtype=torch.float64, device=dev)
    if b == 0:
        return out
    if n > 8:
        _gate(_sc.SIEGEL_FWD, model, n, dev)
    w = _weights(metric, weights, n, dev) if metric == "wsum" else None
    sc_ptr = None
    if scale is not None:
        sc = scale
        if sc.device != dev or sc.dtype != torch.float64:
            sc = sc.detach().to(device=dev, dtype=torch.float64)
        sc_ptr = sc.data_ptr()
    eps = 1e-5 if eps is None else float(eps)
    st = _status_buf(dev)
    tp = triplets.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = lib.sympa_model_forward(tab.data_ptr(), num_rows, n, tp, stride, tp + 8, stride, b,
                                         MODEL_IDS[model], METRIC_IDS[metric], None if w is None else w.data_ptr(),
                                         eps, sc_ptr, float(scale_coef), out.data_ptr(), st.data_ptr(), int(flags), stream)
    else:
        rc = lib.sympa_model_forward(tab.data_ptr(), num_rows, n, tp, stride, tp + 8, stride, b,
                                     MODEL_IDS[model], METRIC_IDS[metric], None if w is None else w.data_ptr(),
                                     eps, sc_ptr, float(scale_coef), out.data_ptr(), st.data_ptr(), int(flags), stream)
    if rc != 0:
        _lib.check(rc)
    if _debug:
        check_status(dev)
    return out


class BatchedForward:
    """Model.forward over a list of batches in ONE C call (C-ABI sympa_model_forward_batches; the loop of
    Runner.evaluate, runner.py:126-137): launch i reads `batches[i]` ([b_i, 2|3] int64) and writes `outs[i]`, on
    streams[i % len(streams)].  The host-side pointer arrays are built once here; `run()` is a single ctypes call
    that enqueues every launch and returns without synchronising.  The caller orders the streams against the
    producers of the inputs / consumers of the outputs (torch stream semantics)."""

    def __init__(self, table, batches, outs, model="upper", metric="riem", weights=None, scale=None, scale_coef=1.0,
                 eps=None, flags=0, streams=None):
        self.lib = _lib.load()
        _need_gpu(table, "table")
        if table.dtype != torch.float64 or table.dim() != 4 or table.shape[1] != 2 or table.shape[2] != table.shape[3]:
            raise ValueError(f"table must be float64 [N,2,n,n], got {tuple(table.shape)} {table.dtype}")
        if not table.is_contiguous():
            raise ValueError("table must be contiguous")
        if len(batches) != len(outs):
            raise ValueError("one output per batch")
        k = len(batches)
        stride = None
        for t, o in zip(batches, outs):
            _need_gpu(t, "triplets"); _need_gpu(o, "out")
            if t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] < 2 or not t.is_contiguous():
                raise TypeError("every batch must be a contiguous int64 [b, >=2] tensor")
            if stride is not None and t.shape[1] != stride:
                raise ValueError("all batches must have the same number of columns")
            stride = t.shape[1]
            if o.dtype != torch.float64 or not o.is_contiguous() or o.numel() < t.shape[0]:
                raise TypeError("every output must be a contiguous float64 tensor of at least b elements")
        self.keep = (table, list(batches), list(outs), weights, scale)
        self.dev = table.device
        self.n = table.shape[2]
        if self.n > 8:
            _gate(_sc.SIEGEL_FWD, model, self.n, self.dev)
        self.num_rows = table.shape[0]
        self.stride = stride or 2
        self.k = k
        self.trip = (ctypes.c_void_p * max(k, 1))(*[t.data_ptr() for t in batches])
        self.b = (ctypes.c_int64 * max(k, 1))(*[t.shape[0] for t in batches])
        self.out = (ctypes.c_void_p * max(k, 1))(*[o.data_ptr() for o in outs])
        self.w = _weights(metric, weights, self.n, self.dev) if metric == "wsum" else None
        self.sc = None
        if scale is not None:
            self.sc = scale if (scale.device == self.dev and scale.dtype == torch.float64) \
                else scale.detach().to(self.dev, torch.float64)
        self.args = (MODEL_IDS[model], METRIC_IDS[metric], None if self.w is None else self.w.data_ptr(),
                     1e-5 if eps is None else float(eps), None if self.sc is None else self.sc.data_ptr(),
                     float(scale_coef))
        self.flags = int(flags)
        self.status = _status_buf(self.dev)
        self.set_streams(streams)

    def set_streams(self, streams):
        if not streams:
            cur = torch.cuda.current_stream(self.dev)
            handle = cur.cuda_stream
            if getattr(self, "_one_stream", None) == handle:
                return                                  # the usual case: same stream as last time, nothing to rebuild
            self._one_stream = handle
            self.streams = [cur]
            self.stream_arr = (ctypes.c_void_p * 1)(handle)
            self._full = None
            return
        self._one_stream = None
        self.streams = list(streams)
        self.stream_arr = (ctypes.c_void_p * len(self.streams))(*[s.cuda_stream for s in self.streams])
        self._full = None

    def run(self, first=0, count=None):
        """Enqueues launches first .. first+count-1 (default: all)."""
        if first == 0 and count is None:
            # the whole list: the argument tuple is converted to C types once (a 20-step timed region is ~100 us; the
            # conversion of 18 Python arguments per call was ~5 us of it)
            full = getattr(self, "_full", None)
            if full is None:
                if self.k == 0:
                    return
                C = ctypes
                full = self._full = (
                    C.c_void_p(self.keep[0].data_ptr()), C.c_int64(self.num_rows), C.c_int(self.n),
                    C.c_void_p(C.addressof(self.trip)), C.c_int64(self.stride), C.c_void_p(C.addressof(self.b)),
                    C.c_int(self.k), C.c_int(self.args[0]), C.c_int(self.args[1]),
                    C.c_void_p(self.args[2]), C.c_double(self.args[3]), C.c_void_p(self.args[4]),
                    C.c_double(self.args[5]), C.c_void_p(C.addressof(self.out)), C.c_void_p(self.status.data_ptr()),
                    C.c_int(self.flags), C.cast(self.stream_arr, C.c_void_p), C.c_int(len(self.streams)))
            rc = self.lib.sympa_model_forward_batches(*full)
            if rc != 0:
                _lib.check(rc)
            if _debug:
                check_status(self.dev)
            return
        count = self.k - first if count is None else count
        if count <= 0:
            return
        if first < 0 or first + count > self.k:
            raise IndexError("batch range outside the prepared list")
        off = first * ctypes.sizeof(ctypes.c_void_p)
        rc = self.lib.sympa_model_forward_batches(
            self.keep[0].data_ptr(), self.num_rows, self.n, ctypes.addressof(self.trip) + off, self.stride,
            ctypes.addressof(self.b) + off, count, self.args[0], self.args[1], self.args[2], self.args[3],
            self.args[4], self.args[5], ctypes.addressof(self.out) + off, self.status.data_ptr(), self.flags,
            self.stream_arr, len(self.streams))
        if rc != 0:
            _lib.check(rc)
        if _debug:
            check_status(self.dev)


def table_changed(param):
    """Tells torch (and with it every cached pack of the table, PackedTable below) that `param` was written IN PLACE by something
    torch cannot see: the HIP optimiser kernels write through raw pointers of `param.data`, and a replayed hipGraph has no Python
    in it at all.  sympa_amd's own optimisers, training steps and the sharded exchange call this after every step."""
    torch.autograd.graph.increment_version(param)


class PackedTable:
    """The packed image of a Siegel table for the INDEXED forward, dims 5..8 (C-ABI sympa_table_pack): one contiguous row per point
    -- the upper triangles of both planes and the inverted Cholesky factor -- made ONCE per table state and reused by every
    batch until the table changes (Runner.evaluate's loop, runner.py:124-135; the N forward calls of the mAP matrix, runner.py:
    142-154; Model.forward, model.py:16-30).

    Validity (round 6) does not depend on the caller's idiom.  The host key -- storage, offset, shape, torch version counter -- sees
    every tracked in-place op and every sympa_amd optimiser step (`table_changed`); a moved key repacks unconditionally.  Writes
    through `.data` (`p.data.add_()` of a torch-1.5 / geoopt optimiser; embeddings.py:36-39 assigns `embeds.data`) move no counter,
    so with `strict` (the default) an UNCHANGED key is not trusted either: C-ABI sympa_table_pack_refresh reads the table once on
    the device (a 64-bit digest, ~10 us for configs[3]'s 46.6 MB), compares it with the digest the pack was made from and
    repacks in the same stream when they differ -- no host synchronisation, and inside a hipGraph capture the pair of kernels
    is recorded, so a replayed graph repacks by itself after an optimiser step.  `strict = False` (or SYMPA_PACK_TRUST_VERSION=1)
    trusts the key alone.
    Memory: the pack is a second table-sized device buffer (0.84 x the table at n = 8) and the pack keeps the table's storage alive
    (a freed table's address cannot return under the same key); `invalidate(release=True)` drops both."""

    strict = os.environ.get("SYMPA_PACK_TRUST_VERSION", "0") in ("", "0")

    @staticmethod
    def supported(table, model):
        return (model in MODEL_IDS and table.is_cuda and table.dtype == torch.float64 and table.dim() == 4 and table.shape[1] == 2
                and table.shape[2] == table.shape[3] and 5 <= table.shape[2] <= 8 and table.is_contiguous())

    def __init__(self, model):
        self.model = model
        self.key = None
        self.pack = None
        self.state = None          # SYMPA_DIGEST_STATE_BYTES of device memory: the digest the pack was made from
        self.repacks = 0           # launches that repack unconditionally (new key); device-decided repacks: `device_repacks()`
        self.refreshes = 0         # digest + guarded-pack launches
        self._src = None
        self._stream = None

    def invalidate(self, release=False):
        self.key = None
        self._seen = None
        if release:
            self.pack = self.state = self._src = None

    def device_repacks(self):
        """How many times the device found the table's bytes changed (or was forced) -- one host read, for tests and profiles."""
        return 0 if self.state is None else int(self.state.view(torch.int32)[7])

    @staticmethod
    def _key_of(table):
        return (table.untyped_storage().data_ptr(), table.storage_offset(), table.shape[0], table.shape[2], table._version,
                table.device)

    def current(self, table, pairs=0):
        """True when the pack is (after the launches this call enqueues) that of `table` as it is now.  Otherwise the table's
        state is remembered and the pack is made the SECOND time the same state is seen: a caller that alternates optimiser
        steps with single forward calls never pays for a pack it would use once, a caller that runs batch after batch over an
        unchanged table packs before its second batch.  (`ensure` packs at once: the list forms -- forward_batches, evaluate
        -- know they have many batches.)  The bounded model packs at FIRST sight when the call has at least 4 pairs per table
        row: its dense kernel factors I - W W^H per pair, and one pack + the packed kernel beats it even for a single call
        (n = 8, 262 144 pairs of 45 500 rows: 27 + 163 us against 245; profiles/r05_packed_forward.txt).
        Inside a stream capture a pack is only used when its buffers exist already (nothing is allocated into the graph's
        pool); the recorded launches then carry the device-side validity check whatever `strict` says."""
        key = self._key_of(table)
        if key == self.key or getattr(self, "_seen", None) == key or \
                (self.model == "bounded" and pairs >= 4 * table.shape[0]):
            return self.ensure(table, _or_none=True) is not None
        self._seen = key
        return False

    # -- the two C calls of the subclasses ---------------------------------------------------------------------------------
    def _need_bytes(self, lib, table):
        return int(lib.sympa_table_pack_bytes(table.shape[0], table.shape[2], MODEL_IDS[self.model]))

    def _refresh(self, lib, table, need, force, stream):
        # no status word for the pack: a point outside the manifold is reported by the PAIRS it enters (its inverted diagonal is
        # stored as NaN), like the reference, whose assertions sit in dist (siegel_manifold.py:64-66) -- a bad row no batch touches
        # raises nothing there either
        return lib.sympa_table_pack_refresh(table.data_ptr(), table.shape[0], table.shape[2], MODEL_IDS[self.model],
                                            self.pack.data_ptr(), need, self.state.data_ptr(), 1 if force else 0, None, stream)

    def _dims_of(self, table):
        return table.shape[2]

    def ensure(self, table, strict=None, _or_none=False):
        key = self._key_of(table)
        same = key == self.key
        capturing = torch.cuda.is_current_stream_capturing()
        strict = self.strict if strict is None else strict
        if same and not strict and not capturing:
            return self
        lib = _lib.load()
        if not type(self).supported(table, self.model):
            raise ValueError(f"{type(self).__name__}: a contiguous float64 device table of the '{self.model}' model at packed dims")
        need = self._need_bytes(lib, table)
        fresh = self.pack is None or self.pack.numel() != need or self.pack.device != table.device
        if fresh:
            if capturing:
                # buffers allocated now would belong to the graph's pool and the host key would describe a pack that does not
                # exist yet (capture runs nothing): the dense kernels serve this capture
                if _or_none:
                    return None
                raise RuntimeError("PackedTable.ensure inside a stream capture needs a pack made before the capture")
            self.pack = torch.empty(need, dtype=torch.uint8, device=table.device)
            self.state = torch.zeros(4096, dtype=torch.uint8, device=table.device)
            self._stream = None
        cur = torch.cuda.current_stream(table.device)
        if self._stream is not None and self._stream != cur and not capturing:
            # the pack was last written on another stream: order this one behind everything enqueued there
            ev = torch.cuda.Event()
            ev.record(self._stream)
            cur.wait_event(ev)
        # (inside a capture the recorded launch must not bake "changed" in: every replay would repack; the digest decides there --
        # the pack depends on the table's bytes alone, so equal bytes under a moved key need no repack either)
        with torch.cuda.device(table.device):
            rc = self._refresh(lib, table, need, (fresh or not same) and not capturing, cur.cuda_stream)
        _lib.check(rc)
        self.refreshes += 1
        if not capturing:          # (a capture runs nothing: the key keeps describing what really is in the buffer)
            if not same:
                self.repacks += 1
            self.key, self._src, self._stream = key, table.untyped_storage(), cur
            self.num_rows, self.n, self.bytes = table.shape[0], self._dims_of(table), need
        return self


def model_forward_packed(packed, triplets, metric="riem", weights=None, scale=None, scale_coef=1.0, eps=None, out=None):
    """Model.forward over a PackedTable (`packed.ensure(table)` first): C-ABI sympa_model_forward_packed."""
    lib = _lib.load()
    _need_gpu(triplets, "triplets")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
        raise TypeError("triplets must be an int64 [b, >=2] tensor (src, dst[, graph_distance])")
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    dev = packed.pack.device
    b = triplets.shape[0]
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    if out is None:
        out = torch.empty(b, dtype=torch.float64, device=dev)
    if b == 0:
        return out
    n, mid = packed.n, MODEL_IDS[packed.model]
    w = _weights(metric, weights, n, dev) if metric == "wsum" else None
    sc_ptr = None
    if scale is not None:
        sc = scale if (scale.device == dev and scale.dtype == torch.float64) else scale.detach().to(device=dev, dtype=torch.float64)
        sc_ptr = sc.data_ptr()
    tp = triplets.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.sympa_model_forward_packed(packed.pack.data_ptr(), packed.bytes, packed.num_rows, n, tp, stride, tp + 8, stride, b,
                                            mid, METRIC_IDS[metric], None if w is None else w.data_ptr(),
                                            1e-5 if eps is None else float(eps), sc_ptr, float(scale_coef), out.data_ptr(),
                                            _status_buf(dev).data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _lib.check(rc)
    if _debug:
        check_status(dev)
    return out


class PackedBatchedForward:
    """BatchedForward over a PackedTable (C-ABI sympa_model_forward_batches_packed): `run()` first makes sure the pack is that of the
    table's current version (one tuple comparison), then ONE C call; up to 32 consecutive batches share a launch."""

    def __init__(self, packed, table, batches, outs, metric="riem", weights=None, scale=None, scale_coef=1.0, eps=None):
        self.lib = _lib.load()
        self.packed, self.table = packed, table
        if len(batches) != len(outs):
            raise ValueError("one output per batch")
        stride = None
        for t, o in zip(batches, outs):
            _need_gpu(t, "triplets"); _need_gpu(o, "out")
            if t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] < 2 or not t.is_contiguous():
                raise TypeError("every batch must be a contiguous int64 [b, >=2] tensor")
            if stride is not None and t.shape[1] != stride:
                raise ValueError("all batches must have the same number of columns")
            stride = t.shape[1]
            if o.dtype != torch.float64 or not o.is_contiguous() or o.numel() < t.shape[0]:
                raise TypeError("every output must be a contiguous float64 tensor of at least b elements")
        self.keep = (list(batches), list(outs), weights, scale)
        self.dev = table.device
        self.n = table.shape[2]
        self.k = len(batches)
        self.stride = stride or 2
        C = ctypes
        self.trip = (C.c_void_p * max(self.k, 1))(*[t.data_ptr() for t in batches])
        self.b = (C.c_int64 * max(self.k, 1))(*[t.shape[0] for t in batches])
        self.out = (C.c_void_p * max(self.k, 1))(*[o.data_ptr() for o in outs])
        self.w = _weights(metric, weights, self.n, self.dev) if metric == "wsum" else None
        self.sc = None
        if scale is not None:
            self.sc = scale if (scale.device == self.dev and scale.dtype == torch.float64) else scale.detach().to(self.dev, torch.float64)
        self.metric = METRIC_IDS[metric]
        self.eps = 1e-5 if eps is None else float(eps)
        self.scale_coef = float(scale_coef)
        self.status = _status_buf(self.dev)

    def set_streams(self, streams):
        if streams:
            raise ValueError("the packed forward runs its launch pairs on ONE stream (the current one)")

    def run(self):
        if self.k == 0:
            return
        pk = self.packed.ensure(self.table)
        C = ctypes
        with torch.cuda.device(self.dev):
            rc = self.lib.sympa_model_forward_batches_packed(
                pk.pack.data_ptr(), pk.bytes, pk.num_rows, self.n, C.addressof(self.trip), self.stride, C.addressof(self.b), self.k,
                MODEL_IDS[pk.model], self.metric, None if self.w is None else self.w.data_ptr(), self.eps,
                None if self.sc is None else self.sc.data_ptr(), self.scale_coef, C.addressof(self.out), self.status.data_ptr(),
                0, torch.cuda.current_stream(self.dev).cuda_stream)
        if rc != 0:
            _lib.check(rc)
        if _debug:
            check_status(self.dev)


def _siegel_bwd_workspace(lib, b, n, model, dev, flags, workspace):
    """The caller-owned scratch of the split Siegel backward (C-ABI sympa_siegel_backward_workspace_bytes; 0 bytes where no kernel
    uses one: dims outside 5..8): `workspace` when given (a persistent uint8 tensor: what a replayed graph wants), else a fresh
    tensor from the caching allocator (stream-ordered; inside a hipGraph capture from the graph's pool).
    SYMPA_SIEGEL_BWD_NO_WORKSPACE=1 (A/B) and FLAG_COOP / FLAG_GENERIC keep the kernels that need none."""
    if n < 5 or n > 8 or (flags & (FLAG_COOP | FLAG_GENERIC)) or os.environ.get("SYMPA_SIEGEL_BWD_NO_WORKSPACE"):
        return None, 0
    if not ((model == "upper" and n >= 7) or (flags & FLAG_SPLIT)):       # where the library would use it (csrc/siegel_bwd.hip)
        return None, 0
    # two launches instead of one: small batches stay with the one-launch kernels unless the caller brings a workspace
    if workspace is None and not (flags & FLAG_SPLIT) and b < int(os.environ.get("SYMPA_SIEGEL_BWD_WORKSPACE_MIN", "1024")):
        return None, 0
    need = int(lib.sympa_siegel_backward_workspace_bytes(int(b), int(n), MODEL_IDS[model]))
    if need <= 0:
        return None, 0
    if workspace is not None:
        if workspace.dtype != torch.uint8 or not workspace.is_cuda or workspace.numel() < need or workspace.data_ptr() % 16:
            raise ValueError(f"workspace: a 16-byte aligned uint8 device tensor of at least {need} bytes")
        return workspace, workspace.numel()
    return torch.empty(need, dtype=torch.uint8, device=dev), need


def siegel_backward_workspace(b, n, model, device, flags=0):
    """A persistent workspace for the backward entries at this (batch, dims, model), or None where the default dispatch would not
    use one (what a replayed graph holds on to: sympa_amd/train_step.py)."""
    lib = _lib.load()
    if n < 5 or n > 8 or (flags & (FLAG_COOP | FLAG_GENERIC)) or os.environ.get("SYMPA_SIEGEL_BWD_NO_WORKSPACE"):
        return None
    if not ((model == "upper" and n >= 7) or (flags & FLAG_SPLIT)):
        return None
    if not (flags & FLAG_SPLIT) and b < int(os.environ.get("SYMPA_SIEGEL_BWD_WORKSPACE_MIN", "1024")):
        return None
    need = int(lib.sympa_siegel_backward_workspace_bytes(int(b), int(n), MODEL_IDS[model]))
    return torch.empty(need, dtype=torch.uint8, device=device) if need > 0 else None


def siegel_backward_workspace_bytes(b, n, model="upper"):
    """Bytes of scratch the split backward of dims 5..8 wants for a batch of b pairs (0: no kernel uses one)."""
    return int(_lib.load().sympa_siegel_backward_workspace_bytes(int(b), int(n), MODEL_IDS[model]))


def siegel_dist_backward(z1, z2, grad_out, model="upper", metric="riem", weights=None, eps=None, flags=0, workspace=None):
    """Backward of manifold.dist for pre-gathered points (C-ABI sympa_siegel_dist_bwd).
    Returns (grad_z1, grad_z2, grad_weights or None): what torch autograd produces through the
    reference's dist (runner.py:105 over siegel_manifold.py:41-72)."""
    lib = _lib.load()
    _need_gpu(z1, "z1"); _need_gpu(z2, "z2"); _need_gpu(grad_out, "grad_out")
    z1 = z1.detach().contiguous()
    z2 = z2.detach().contiguous()
    go = grad_out.detach().to(torch.float64).contiguous()
    b, _, n, _ = z1.shape
    g1 = torch.empty_like(z1)
    _gate(_sc.SIEGEL_BWD, model, n, z1.device)
    g2 = torch.empty_like(z2)
    w = _weights(metric, weights, n, z1.device)
    gw = torch.zeros(n, dtype=torch.float64, device=z1.device) if metric == "wsum" else None
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(z1.device)
    if b > 0:
        ws, ws_bytes = _siegel_bwd_workspace(lib, b, n, model, z1.device, int(flags), workspace)
        with torch.cuda.device(z1.device):
            rc = lib.sympa_siegel_dist_bwd(_ptr(z1), _ptr(z2), _ptr(go), b, n, MODEL_IDS[model], METRIC_IDS[metric],
                                           _ptr(w), eps, _ptr(g1), _ptr(g2), _ptr(gw), _ptr(st), _ptr(ws), ws_bytes, int(flags),
                                           _stream())
        _lib.check(rc)
    if _debug:
        check_status(z1.device)
    return g1, g2, gw


def model_backward(table, triplets, grad_out, model="upper", metric="riem", weights=None, scale=None,
                   scale_coef=1.0, eps=None, grad_table=None, flags=0, workspace=None):
    """Backward of the fused Model.forward (C-ABI sympa_model_backward): scatter-adds the two gradient
    rows of every pair into a dense [N,2,n,n] gradient (created zeroed unless `grad_table` is given, in
    which case it accumulates), returns (grad_table, grad_weights or None, grad_scale or None)."""
    lib = _lib.load()
    _need_gpu(table, "table"); _need_gpu(triplets, "triplets"); _need_gpu(grad_out, "grad_out")
    tab = table.detach()
    if not tab.is_contiguous():
        tab = tab.contiguous()
    num_rows, _, n, _ = tab.shape
    b = triplets.shape[0]
    _gate(_sc.SIEGEL_BWD, model, n, tab.device)
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    go = grad_out.detach().to(torch.float64).contiguous()
    if grad_table is None:
        grad_table = torch.zeros_like(tab)
    w = _weights(metric, weights, n, tab.device)
    gw = torch.zeros(n, dtype=torch.float64, device=tab.device) if metric == "wsum" else None
    sc = gs = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=tab.device, dtype=torch.float64).contiguous()
        gs = torch.zeros(1, dtype=torch.float64, device=tab.device)
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(tab.device)
    if b > 0:
        src_ptr = ctypes.c_void_p(triplets.data_ptr())
        dst_ptr = ctypes.c_void_p(triplets.data_ptr() + 8)
        ws, ws_bytes = _siegel_bwd_workspace(lib, b, n, model, tab.device, int(flags), workspace)
        with torch.cuda.device(tab.device):
            rc = lib.sympa_model_backward(_ptr(tab), num_rows, n, src_ptr, stride, dst_ptr, stride, b,
                                          MODEL_IDS[model], METRIC_IDS[metric], _ptr(w), eps, _ptr(sc),
                                          float(scale_coef), _ptr(go), _ptr(grad_table), _ptr(gw), _ptr(gs), None,
                                          _ptr(st), _ptr(ws), ws_bytes, int(flags), _stream())
        _lib.check(rc)
    if _debug:
        check_status(tab.device)
    return grad_table, gw, gs


def model_loss_backward(table, triplets, graph_dist, grad_table, loss, model="upper", metric="riem", weights=None,
                        grad_weights=None, scale=None, grad_scale=None, scale_coef=1.0, loss_scale=1.0, eps=None, flags=0,
                        workspace=None):
    """Fused training step (C-ABI sympa_model_loss_backward): distances + AverageDistortionLoss + all
    gradients in one kernel.  `grad_table` [N,2,n,n], `loss` [1] (and `grad_weights` [n], `grad_scale` [1]
    when given) are ACCUMULATED into, like autograd's .grad."""
    lib = _lib.load()
    tab = table if table.is_contiguous() else table.contiguous()
    num_rows, n = tab.shape[0], tab.shape[2]
    b = triplets.shape[0]
    if b == 0:
        return loss
    if n >= 7:
        _gate(_sc.SIEGEL_BWD, model, n, tab.device)
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    dev = tab.device
    if not (tab.is_cuda and triplets.is_cuda and graph_dist.is_cuda):
        _need_gpu(tab, "table"); _need_gpu(triplets, "triplets"); _need_gpu(graph_dist, "graph_dist")
    gd = graph_dist if (graph_dist.dtype == torch.float64 and graph_dist.is_contiguous()) \
        else graph_dist.to(torch.float64).contiguous()
    w = _weights(metric, weights, n, dev) if metric == "wsum" else None
    sc_ptr = None
    if scale is not None:
        sc = scale if (scale.device == dev and scale.dtype == torch.float64) else scale.detach().to(dev, torch.float64)
        sc_ptr = sc.data_ptr()
    eps = 1e-5 if eps is None else float(eps)
    st = _status_buf(dev)
    tp = triplets.data_ptr()
    ws, ws_bytes = _siegel_bwd_workspace(lib, b, n, model, dev, int(flags), workspace)
    with torch.cuda.device(dev):
        rc = lib.sympa_model_loss_backward(
            tab.data_ptr(), num_rows, n, tp, stride, tp + 8, stride, gd.data_ptr(), b, MODEL_IDS[model],
            METRIC_IDS[metric], None if w is None else w.data_ptr(), eps, sc_ptr, float(scale_coef), float(loss_scale),
            loss.data_ptr(), grad_table.data_ptr(), None if grad_weights is None else grad_weights.data_ptr(),
            None if grad_scale is None else grad_scale.data_ptr(), None, st.data_ptr(),
            None if ws is None else ws.data_ptr(), ws_bytes, int(flags), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _lib.check(rc)
    if _debug:
        check_status(dev)
    return loss


def model_loss_backward_rows(table, triplets, graph_dist, grad_rows, loss, model="upper", metric="riem", weights=None,
                             grad_weights=None, scale=None, grad_scale=None, scale_coef=1.0, loss_scale=1.0, eps=None,
                             flags=0, workspace=None):
    """The fused training step with the table gradient left per pair (C-ABI sympa_model_loss_backward_rows):
    `grad_rows` [2b, 2, n, n] is WRITTEN -- rows [0, b) belong to triplets[:, 0], rows [b, 2b) to triplets[:, 1];
    loss / grad_weights / grad_scale are accumulated as in model_loss_backward."""
    lib = _lib.load()
    tab = table if table.is_contiguous() else table.contiguous()
    num_rows, n = tab.shape[0], tab.shape[2]
    b = triplets.shape[0]
    if b == 0:
        return loss
    if n >= 7:
        _gate(_sc.SIEGEL_BWD, model, n, tab.device)
    _need_gpu(tab, "table"); _need_gpu(triplets, "triplets"); _need_gpu(graph_dist, "graph_dist"); _need_gpu(grad_rows, "grad_rows")
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    if grad_rows.dtype != torch.float64 or not grad_rows.is_contiguous() or grad_rows.numel() < 2 * b * 2 * n * n:
        raise ValueError("grad_rows must be a contiguous float64 tensor of at least [2b, 2, n, n]")
    dev = tab.device
    gd = graph_dist if (graph_dist.dtype == torch.float64 and graph_dist.is_contiguous()) \
        else graph_dist.to(torch.float64).contiguous()
    w = _weights(metric, weights, n, dev) if metric == "wsum" else None
    sc_ptr = None
    if scale is not None:
        sc = scale if (scale.device == dev and scale.dtype == torch.float64) else scale.detach().to(dev, torch.float64)
        sc_ptr = sc.data_ptr()
    eps = 1e-5 if eps is None else float(eps)
    st = _status_buf(dev)
    tp = triplets.data_ptr()
    rowbytes = 2 * n * n * 8
    ws, ws_bytes = _siegel_bwd_workspace(lib, b, n, model, dev, int(flags), workspace)
    with torch.cuda.device(dev):
        rc = lib.sympa_model_loss_backward_rows(
            tab.data_ptr(), num_rows, n, tp, stride, tp + 8, stride, gd.data_ptr(), b, MODEL_IDS[model],
            METRIC_IDS[metric], None if w is None else w.data_ptr(), eps, sc_ptr, float(scale_coef), float(loss_scale),
            loss.data_ptr(), grad_rows.data_ptr(), grad_rows.data_ptr() + b * rowbytes,
            None if grad_weights is None else grad_weights.data_ptr(),
            None if grad_scale is None else grad_scale.data_ptr(), None, st.data_ptr(),
            None if ws is None else ws.data_ptr(), ws_bytes, int(flags), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _lib.check(rc)
    if _debug:
        check_status(dev)
    return loss


def scatter_add_rows_(grad_table, rows, idx, alpha=1.0):
    """grad_table[idx[r]] += alpha * rows[r]  (C-ABI sympa_scatter_add_rows; fp64 atomics, whole contiguous rows)."""
    lib = _lib.load()
    _need_gpu(grad_table, "grad_table"); _need_gpu(rows, "rows"); _need_gpu(idx, "idx")
    if grad_table.dtype != torch.float64 or rows.dtype != torch.float64 or idx.dtype != torch.int64:
        raise TypeError("scatter_add_rows_ needs float64 rows / table and int64 indices")
    if not (grad_table.is_contiguous() and rows.is_contiguous() and idx.dim() == 1):
        raise ValueError("contiguous tensors and a 1-d index list expected")
    n = grad_table.shape[-1]
    count = idx.shape[0]
    if rows.numel() != count * 2 * n * n:
        raise ValueError("rows must hold one [2, n, n] gradient row per index")
    st = _status_buf(grad_table.device)
    with torch.cuda.device(grad_table.device):
        rc = lib.sympa_scatter_add_rows(rows.data_ptr(), idx.data_ptr(), idx.stride(0), count, n, grad_table.shape[0],
                                        float(alpha), grad_table.data_ptr(), st.data_ptr(), _stream())
    _lib.check(rc)
    return grad_table


# ---------------------------------------------------------------------------------------------------
# optimiser-side manifold operations over table rows (C-ABI sympa_egrad2rgrad / sympa_projx / sympa_rsgd_step)
# ---------------------------------------------------------------------------------------------------
def _rows(t, name):
    _need_gpu(t, name)
    if t.dtype != torch.float64 or t.dim() != 4 or t.shape[1] != 2 or t.shape[2] != t.shape[3]:
        raise ValueError(f"{name} must be a float64 [b,2,n,n] tensor, got {tuple(t.shape)} {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def egrad2rgrad(z, u, model):
    lib = _lib.load()
    z, u = _rows(z.detach(), "z"), _rows(u.detach(), "u")
    _gate(_sc.SIEGEL_TABLE, model, z.shape[2], z.device)
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        rc = lib.sympa_egrad2rgrad(z.data_ptr(), u.data_ptr(), z.shape[0], z.shape[2], MODEL_IDS[model], out.data_ptr(),
                                   _stream())
    _lib.check(rc)
    return out


def tangent_sqnorm(z, u, model):
    """inner(z, u, u) per row (C-ABI sympa_tangent_sqnorm; upper_half.py:68-91 / bounded_domain.py:86-116) -> [b]."""
    lib = _lib.load()
    z, u = _rows(z.detach(), "z"), _rows(u.detach(), "u")
    _gate(_sc.SIEGEL_TABLE, model, z.shape[2], z.device)
    out = torch.empty(z.shape[0], dtype=torch.float64, device=z.device)
    st = _status_buf(z.device)
    with torch.cuda.device(z.device):
        rc = lib.sympa_tangent_sqnorm(z.data_ptr(), u.data_ptr(), z.shape[0], z.shape[2], MODEL_IDS[model], out.data_ptr(),
                                      st.data_ptr(), _stream())
    _lib.check(rc)
    return out


def _outside_word(n, dev):
    """The scratch word the dims >= 7 projection is gated on (C-ABI `outside_word`): the CALLER's, one per call in flight.
    A fresh one-element tensor per call: the caching allocator hands it out stream-ordered on the current stream (inside a
    hipGraph capture: from the graph's pool), so calls on different streams, devices or threads never share a word and the
    library needs no state of its own."""
    return torch.empty(1, dtype=torch.int32, device=dev) if n >= 7 else None


def projx(z, model, eps=None, counter=None):
    """Returns projx(z); `counter` (int32[1] device tensor) += rows that were moved."""
    lib = _lib.load()
    z = _rows(z.detach(), "z")
    _gate(_sc.SIEGEL_TABLE, model, z.shape[2], z.device)
    out = torch.empty_like(z)
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(z.device)
    word = _outside_word(z.shape[2], z.device)
    with torch.cuda.device(z.device):
        rc = lib.sympa_projx(z.data_ptr(), z.shape[0], z.shape[2], MODEL_IDS[model], eps, out.data_ptr(),
                             None if counter is None else counter.data_ptr(), st.data_ptr(),
                             None if word is None else word.data_ptr(), _stream())
    _lib.check(rc)
    return out


def sgd_step_clipped_(param, grad, lr, weight_decay=0.0, clip_sqnorm=None, max_norm=None):
    """param <- param - lr * (coef * grad + wd * param) in one kernel (C-ABI sympa_sgd_step_clipped): the step of a
    parameter without a manifold, with the gradient clip folded in like the table step."""
    lib = _lib.load()
    _need_gpu(param, "param"); _need_gpu(grad, "grad")
    if param.dtype != torch.float64 or grad.dtype != torch.float64 or not param.is_contiguous() or not grad.is_contiguous():
        raise ValueError("param and grad must be contiguous float64 tensors")
    with torch.cuda.device(param.device):
        rc = lib.sympa_sgd_step_clipped(param.data_ptr(), grad.data_ptr(), param.numel(), float(lr), float(weight_decay),
                                        None if clip_sqnorm is None else clip_sqnorm.data_ptr(),
                                        float(max_norm) if max_norm is not None else 0.0, _stream())
    _lib.check(rc)
    return param


def sqnorm_accum_(x, acc):
    """acc[0] += sum(x^2)  (C-ABI sympa_sqnorm_accum): the squared total norm of clip_grad_norm_, on the device."""
    lib = _lib.load()
    _need_gpu(x, "x")
    if x.dtype != torch.float64 or acc.dtype != torch.float64 or not x.is_contiguous():
        raise TypeError("sqnorm_accum_ needs contiguous float64 tensors")
    with torch.cuda.device(x.device):
        rc = lib.sympa_sqnorm_accum(x.data_ptr(), x.numel(), acc.data_ptr(), _stream())
    _lib.check(rc)
    return acc


def rsgd_step_(table, grad, model, lr, weight_decay=0.0, eps=None, counter=None, clip_sqnorm=None, max_norm=None):
    """In-place RiemannianSGD step over the whole table (one kernel).  With clip_sqnorm (1-element device tensor
    holding the squared total gradient norm) and max_norm, the gradient rows are scaled like clip_grad_norm_ does."""
    lib = _lib.load()
    _need_gpu(table, "table")
    if not table.is_contiguous():
        raise ValueError("table must be contiguous for the in-place step")
    grad = _rows(grad.detach(), "grad")
    if table.dtype != torch.float64 or table.shape != grad.shape:
        raise ValueError(f"table must be a float64 tensor of the gradient's shape {tuple(grad.shape)}, got "
                         f"{tuple(table.shape)} {table.dtype}")
    if table.shape[2] >= 7:
        _gate(_sc.SIEGEL_TABLE, model, table.shape[2], table.device)
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(table.device)
    word = _outside_word(table.shape[2], table.device)
    wp = None if word is None else word.data_ptr()
    with torch.cuda.device(table.device):
        if clip_sqnorm is not None:
            rc = lib.sympa_rsgd_step_clipped(table.data_ptr(), grad.data_ptr(), table.shape[0], table.shape[2],
                                             MODEL_IDS[model], float(lr), float(weight_decay), eps,
                                             clip_sqnorm.data_ptr(), float(max_norm),
                                             None if counter is None else counter.data_ptr(), st.data_ptr(), wp, _stream())
        else:
            rc = lib.sympa_rsgd_step(table.data_ptr(), grad.data_ptr(), table.shape[0], table.shape[2], MODEL_IDS[model],
                                     float(lr), float(weight_decay), eps, None if counter is None else counter.data_ptr(),
                                     st.data_ptr(), wp, _stream())
    _lib.check(rc)
    return table


RADAM_FUSED_MAX_DIMS = 6


def radam_step_(table, grad, exp_avg, exp_avg_sq, bias_pows, model, lr, betas=(0.9, 0.999), eps_adam=1e-8, weight_decay=0.0,
                eps=None, counter=None):
    """In-place RiemannianAdam step over the whole table as ONE kernel (C-ABI sympa_radam_step, dims <= 6): table / grad /
    exp_avg [N,2,n,n], exp_avg_sq [N], bias_pows = device tensor (beta1^t, beta2^t) of this step."""
    lib = _lib.load()
    _need_gpu(table, "table")
    grad = _rows(grad.detach(), "grad")
    for name, t, shape in (("table", table, grad.shape), ("exp_avg", exp_avg, grad.shape), ("exp_avg_sq", exp_avg_sq, grad.shape[:1]),
                           ("bias_pows", bias_pows, (2,))):
        if t.dtype != torch.float64 or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != table.device:
            raise ValueError(f"{name} must be a contiguous float64 tensor of shape {tuple(shape)} on {table.device}, got "
                             f"{tuple(t.shape)} {t.dtype} on {t.device}")
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(table.device)
    with torch.cuda.device(table.device):
        rc = lib.sympa_radam_step(table.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), table.shape[0],
                                  table.shape[2], MODEL_IDS[model], float(lr), float(betas[0]), float(betas[1]), float(eps_adam),
                                  float(weight_decay), bias_pows.data_ptr(), eps,
                                  None if counter is None else counter.data_ptr(), st.data_ptr(), _stream())
    _lib.check(rc)
    return table


def model_train_backward(table, triplets, graph_dist, batch, loss, model="upper", metric="riem", weights=None,
                         grad_weights=None, scale=None, grad_scale=None, scale_coef=1.0, loss_scale=1.0, grad_table=None,
                         grad_rows=None, step_counter=None, wave_partials=None, eps=None, flags=0, workspace=None):
    """The backward half of a training step for replayed graphs (C-ABI sympa_model_train_backward; wave_partials: dims <= 8): pairs
    [c * batch, (c + 1) * batch) of `triplets` [T, >=2] int64 / `graph_dist` [T] fp64, c = step_counter[0] (device int64; None:
    0).  grad_table: atomic scatter into the dense gradient; grad_rows [2 batch, 2, n, n]: per-pair rows (written).
    wave_partials [ceil(batch / 64), 2 + n]: deterministic mode, see segment_sum_rows_."""
    lib = _lib.load()
    _need_gpu(table, "table"); _need_gpu(triplets, "triplets"); _need_gpu(graph_dist, "graph_dist")
    if table.dtype != torch.float64 or not table.is_contiguous() or table.dim() != 4:
        raise ValueError("table must be a contiguous float64 [N,2,n,n] tensor")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2 or triplets.stride(1) != 1:
        raise TypeError("triplets must be an int64 [T, >=2] tensor with unit column stride")
    if graph_dist.dtype != torch.float64 or not graph_dist.is_contiguous():
        raise TypeError("graph_dist must be a contiguous float64 tensor")
    n = table.shape[2]
    dev = table.device
    if step_counter is None and triplets.shape[0] < batch:
        raise ValueError("fewer triplets than the batch size")
    if grad_rows is not None and (grad_rows.dtype != torch.float64 or not grad_rows.is_contiguous()
                                  or grad_rows.numel() < 2 * batch * 2 * n * n):
        raise ValueError("grad_rows must be a contiguous float64 tensor of at least [2 batch, 2, n, n]")
    if wave_partials is not None and (wave_partials.dtype != torch.float64 or not wave_partials.is_contiguous()
                                      or wave_partials.numel() < ((batch + 63) // 64) * (2 + n)):
        raise ValueError("wave_partials must be a contiguous float64 tensor of at least [ceil(batch / 64), 2 + n]")
    w = _weights(metric, weights, n, dev) if metric == "wsum" else None
    sc_ptr = None
    if scale is not None:
        sc = scale if (scale.device == dev and scale.dtype == torch.float64) else scale.detach().to(dev, torch.float64)
        sc_ptr = sc.data_ptr()
    st = _status_buf(dev)
    tp = triplets.data_ptr()
    stride = triplets.stride(0)
    ws, ws_bytes = _siegel_bwd_workspace(lib, int(batch), n, model, dev, int(flags), workspace)
    with torch.cuda.device(dev):
        rc = lib.sympa_model_train_backward(
            table.data_ptr(), table.shape[0], n, tp, stride, tp + 8, stride, graph_dist.data_ptr(), int(batch),
            None if step_counter is None else step_counter.data_ptr(), MODEL_IDS[model], METRIC_IDS[metric],
            None if w is None else w.data_ptr(), 1e-5 if eps is None else float(eps), sc_ptr, float(scale_coef),
            float(loss_scale), loss.data_ptr(), None if grad_table is None else grad_table.data_ptr(),
            None if grad_rows is None else grad_rows.data_ptr(), None if grad_weights is None else grad_weights.data_ptr(),
            None if grad_scale is None else grad_scale.data_ptr(),
            None if wave_partials is None else wave_partials.data_ptr(), st.data_ptr(),
            None if ws is None else ws.data_ptr(), ws_bytes, int(flags), _stream())
    if rc != 0:
        _lib.check(rc)
    return loss


def sorted_slots(row_index_lists, num_rows, merged_src=0):
    """The lists the deterministic gradient accumulation adds in (segment_sum_rows_): for every batch s, the slots
    0 .. 2b-1 of its per-pair gradient rows (slot k < b: row src[k]; slot b + k: row dst[k]) sorted by table row -- stable,
    so inside a table row the slots stay in batch order -- and the CSR pointers of the table rows into that list.
    row_index_lists: int64 [steps, 2b] (table row of every slot).  Returns (order int32 [steps, 2b], rowptr int32
    [steps, num_rows + 1]).  Torch ops on the device the indices live on: one sort per EPOCH, not per step.
    merged_src = b: the rows were written with FLAG_MERGE_SRC -- of every run of equal source ids inside a wave (64 consecutive
    pairs) only the LAST slot holds a row (the run's sum): the other source slots are left out of the lists."""
    keys = row_index_lists
    if keys.dim() == 1:
        keys = keys.unsqueeze(0)
    if merged_src:
        b = int(merged_src)
        src = keys[:, :b]
        pos = torch.arange(b, device=keys.device)
        ends = torch.ones_like(src, dtype=torch.bool)
        ends[:, :-1] = (src[:, 1:] != src[:, :-1]) | ((pos[:-1] & 63) == 63)
        keys = torch.cat((torch.where(ends, src, torch.full_like(src, num_rows)), keys[:, b:]), dim=1)    # (num_rows: beyond rowptr)
    sorted_rows, order = torch.sort(keys, dim=1, stable=True)
    bounds = torch.arange(num_rows + 1, device=keys.device, dtype=keys.dtype).unsqueeze(0).expand(keys.shape[0], -1)
    rowptr = torch.searchsorted(sorted_rows.contiguous(), bounds.contiguous(), right=False)
    return order.to(torch.int32).contiguous(), rowptr.to(torch.int32).contiguous()


def segment_sum_rows_(grad_table, rows, order, rowptr, alpha=1.0, accumulate=False, step_counter=None,
                      wave_partials=None, num_waves=0, partial_stride=0, loss=None, grad_scale=None, grad_weights=None,
                      sq_partials=None):
    """grad_table[r] (= | +=) alpha * sum of rows[order[p]], p in [rowptr[r], rowptr[r + 1]), added in list order: the
    deterministic counterpart of scatter_add_rows_ (C-ABI sympa_segment_sum_rows).  order / rowptr from sorted_slots();
    with step_counter the lists of batch step_counter[0] are used.  wave_partials: the per-wave sums model_train_backward
    left; they are added to loss / grad_scale / grad_weights in a fixed order."""
    lib = _lib.load()
    _need_gpu(grad_table, "grad_table"); _need_gpu(rows, "rows"); _need_gpu(order, "order"); _need_gpu(rowptr, "rowptr")
    if grad_table.dtype != torch.float64 or rows.dtype != torch.float64 or not grad_table.is_contiguous() or not rows.is_contiguous():
        raise TypeError("contiguous float64 gradient table and rows expected")
    if order.dtype != torch.int32 or rowptr.dtype != torch.int32 or not order.is_contiguous() or not rowptr.is_contiguous():
        raise TypeError("order and rowptr must be contiguous int32 tensors (sorted_slots())")
    num_rows = grad_table.shape[0]
    rowd = grad_table[0].numel()
    if rowptr.shape[-1] != num_rows + 1:
        raise ValueError("rowptr must have num_rows + 1 entries per batch")
    if step_counter is None and rows.numel() < order.shape[-1] * rowd:
        raise ValueError("rows must hold one gradient row per slot")
    nw = 0 if grad_weights is None else grad_weights.numel()
    with torch.cuda.device(grad_table.device):
        rc = lib.sympa_segment_sum_rows(
            rows.data_ptr(), order.data_ptr(), rowptr.data_ptr(), num_rows, rowd, order.shape[-1],
            None if step_counter is None else step_counter.data_ptr(), float(alpha), 1 if accumulate else 0,
            grad_table.data_ptr(), None if wave_partials is None else wave_partials.data_ptr(), int(num_waves),
            int(partial_stride), nw,
            None if loss is None else loss.data_ptr(), None if grad_scale is None else grad_scale.data_ptr(),
            None if grad_weights is None else grad_weights.data_ptr(),
            None if sq_partials is None else sq_partials.data_ptr(), _stream())
    _lib.check(rc)
    return grad_table


def segment_sum_partials(grad_table):
    """Length of the sq_partials buffer segment_sum_rows_ fills for this gradient table."""
    return int(_lib.load().sympa_segment_sum_partials(grad_table.shape[0], grad_table[0].numel()))


class FusedStep:
    """clip_grad_norm_ + RiemannianSGD step of the table + plain SGD step of up to two small parameters (the scale, the
    wsum weights) + zero_grad as ONE launch (C-ABI sympa_rsgd_step_fused; runner.py:113-118); with `adam`: the same for
    RiemannianAdam / Adam (sympa_radam_step_fused).  Built once per (table,
    gradient, parameters): the pointer arrays and the zeroed workspace live here, `run()` is one ctypes call.  Raises
    SympaHipError with code SYMPA_ERR_UNSUPPORTED_DIMS when the table does not qualify (dims > 6 or more row blocks than
    CUs): the caller then keeps the separate kernels.  `supported()` tells in advance."""

    @staticmethod
    def supported(table):
        if not (table.is_cuda and table.dtype == torch.float64 and table.dim() == 4 and table.shape[2] <= 6):
            return False
        cus = torch.cuda.get_device_properties(table.device).multi_processor_count
        return (table.shape[0] + 255) // 256 <= cus

    def __init__(self, table, grad, model, extras=(), counter=None, projected=None, zero_grads=True, sq_partials=None,
                 adam=None):
        """adam: None (RiemannianSGD) or a dict(exp_avg, exp_avg_sq, bias_pows, betas, eps, extras=[(exp_avg, exp_avg_sq,
        bias_pows), ...]) of the RiemannianAdam state tensors (C-ABI sympa_radam_step_fused): the kernel advances the powers."""
        self.lib = _lib.load()
        _need_gpu(table, "table"); _need_gpu(grad, "grad")
        if table.dtype != torch.float64 or grad.dtype != torch.float64 or table.shape != grad.shape \
                or not table.is_contiguous() or not grad.is_contiguous() or table.dim() != 4 or table.shape[1] != 2:
            raise ValueError("table and grad must be contiguous float64 [N,2,n,n] tensors of the same shape")
        if len(extras) > 2:
            raise ValueError("at most two plain parameters")
        for p, g in extras:
            _need_gpu(p, "extra parameter"); _need_gpu(g, "extra gradient")
            if p.dtype != torch.float64 or g.dtype != torch.float64 or p.numel() != g.numel() or not (1 <= p.numel() <= 64) \
                    or not p.is_contiguous() or not g.is_contiguous():
                raise ValueError("plain parameters: contiguous float64, 1..64 elements, gradient of the same size")
        self.keep = (table, grad, list(extras), counter, projected)
        self.adam = None
        if adam is not None:
            ax = list(adam.get("extras", ()))
            if len(ax) != len(extras):
                raise ValueError("one (exp_avg, exp_avg_sq, bias_pows) triple per plain parameter")
            def _chk(t, shape, what):
                if t.dtype != torch.float64 or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != table.device:
                    raise ValueError(f"{what} must be a contiguous float64 tensor of shape {tuple(shape)} on {table.device}")
            _chk(adam["exp_avg"], table.shape, "exp_avg")
            _chk(adam["exp_avg_sq"], table.shape[:1], "exp_avg_sq")
            _chk(adam["bias_pows"], (2,), "bias_pows")
            for (p, _), (m_, v_, w_) in zip(extras, ax):
                _chk(m_, p.shape, "plain exp_avg"); _chk(v_, p.shape, "plain exp_avg_sq"); _chk(w_, (2,), "plain bias_pows")
            k_ = max(len(ax), 1)
            self.adam = dict(adam, extras=ax,
                             xm=(ctypes.c_void_p * k_)(*[t[0].data_ptr() for t in ax]),
                             xv=(ctypes.c_void_p * k_)(*[t[1].data_ptr() for t in ax]),
                             xw=(ctypes.c_void_p * k_)(*[t[2].data_ptr() for t in ax]))
        self.dev = table.device
        self.model = MODEL_IDS[model]
        k = len(extras)
        self.k = k
        self.xp = (ctypes.c_void_p * max(k, 1))(*[p.data_ptr() for p, _ in extras])
        self.xg = (ctypes.c_void_p * max(k, 1))(*[g.data_ptr() for _, g in extras])
        self.xn = (ctypes.c_int * max(k, 1))(*[p.numel() for p, _ in extras])
        need = self.lib.sympa_rsgd_step_fused_workspace_bytes(table.shape[0])
        self.ws = torch.zeros(need // 8, dtype=torch.float64, device=self.dev)
        self.zero_grads = 1 if zero_grads else 0
        self.status = _status_buf(self.dev)
        # squared-norm partials left by segment_sum_rows_ (deterministic step): no pass over the gradient, no grid barrier
        self.sq_partials = sq_partials
        if sq_partials is not None and (sq_partials.dtype != torch.float64 or not sq_partials.is_cuda or not sq_partials.is_contiguous()):
            raise ValueError("sq_partials must be a contiguous float64 device tensor")

    def run(self, lr, weight_decay=0.0, max_norm=None, extra_lr=(), extra_weight_decay=(), eps=None):
        table, grad, extras, counter, projected = self.keep
        xlr = (ctypes.c_double * max(self.k, 1))(*[float(x) for x in extra_lr])
        xwd = (ctypes.c_double * max(self.k, 1))(*[float(x) for x in extra_weight_decay])
        if len(extra_lr) != self.k or len(extra_weight_decay) != self.k:
            raise ValueError("one learning rate and weight decay per plain parameter")
        if self.adam is not None:
            ad = self.adam
            with torch.cuda.device(self.dev):
                rc = self.lib.sympa_radam_step_fused(
                    table.data_ptr(), grad.data_ptr(), ad["exp_avg"].data_ptr(), ad["exp_avg_sq"].data_ptr(),
                    ad["bias_pows"].data_ptr(), table.shape[0], table.shape[2], self.model, float(lr), float(ad["betas"][0]),
                    float(ad["betas"][1]), float(ad["eps"]), float(weight_decay), EPS[torch.float64] if eps is None else float(eps),
                    float(max_norm) if max_norm is not None else 0.0, self.zero_grads, self.xp, self.xg, ad["xm"], ad["xv"],
                    ad["xw"], self.xn, xlr, xwd, self.k, self.ws.data_ptr(), self.ws.numel() * 8,
                    None if (self.sq_partials is None or max_norm is None) else self.sq_partials.data_ptr(),
                    0 if self.sq_partials is None else self.sq_partials.numel(), None if counter is None else counter.data_ptr(),
                    None if projected is None else projected.data_ptr(), self.status.data_ptr(), _stream())
            _lib.check(rc)
            return
        with torch.cuda.device(self.dev):
            rc = self.lib.sympa_rsgd_step_fused(
                table.data_ptr(), grad.data_ptr(), table.shape[0], table.shape[2], self.model, float(lr),
                float(weight_decay), EPS[torch.float64] if eps is None else float(eps),
                float(max_norm) if max_norm is not None else 0.0, self.zero_grads, self.xp, self.xg, self.xn, xlr, xwd, self.k,
                self.ws.data_ptr(), self.ws.numel() * 8,
                None if (self.sq_partials is None or max_norm is None) else self.sq_partials.data_ptr(),
                0 if self.sq_partials is None else self.sq_partials.numel(), None if counter is None else counter.data_ptr(),
                None if projected is None else projected.data_ptr(), self.status.data_ptr(), _stream())
        _lib.check(rc)


def all_pairs_dist(table, model="upper", metric="riem", weights=None, scale=None, scale_coef=1.0, row_begin=0,
                   row_count=None, eps=None, out=None, packed=None, workspace=None, flags=0):
    """Rows [row_begin, row_begin + row_count) of the N x N distance matrix (C-ABI sympa_all_pairs_dist;
    reference Runner.build_distance_matrix, runner.py:142-154).  Returns [row_count, N] fp64."""
    lib = _lib.load()
    _need_gpu(table, "table")
    tab = table.detach()
    tab = tab if tab.is_contiguous() else tab.contiguous()
    num_rows, n = tab.shape[0], tab.shape[2]
    if n > 8:
        _gate(_sc.SIEGEL_FWD, model, n, tab.device)
    row_count = num_rows - row_begin if row_count is None else int(row_count)
    if out is None:
        out = torch.empty(row_count, num_rows, dtype=torch.float64, device=tab.device)
    w = _weights(metric, weights, n, tab.device) if metric == "wsum" else None
    sc = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=tab.device, dtype=torch.float64).contiguous()
    eps = EPS[torch.float64] if eps is None else float(eps)
    st = _status_buf(tab.device)
    # per-point factor reuse (C-ABI sympa_all_pairs_dist_packed) where the build has it; packed=False forces the
    # pairwise kernel in index-free mode (the tests compare the two)
    need = lib.sympa_all_pairs_workspace_bytes(num_rows, n, MODEL_IDS[model]) if model in MODEL_IDS else 0
    if packed is None:
        packed = need > 0
    if packed:
        if need <= 0:
            raise _lib.SympaHipError(f"no packed all-pairs kernel for dims {n}")
        if workspace is None:
            workspace = torch.empty(need // 8, dtype=torch.float64, device=tab.device)
        elif workspace.numel() * workspace.element_size() < need or not workspace.is_cuda:
            raise ValueError("workspace too small")
        with torch.cuda.device(tab.device):
            rc = lib.sympa_all_pairs_dist_packed(tab.data_ptr(), num_rows, n, int(row_begin), row_count,
                                                 MODEL_IDS[model], METRIC_IDS[metric],
                                                 None if w is None else w.data_ptr(), eps,
                                                 None if sc is None else sc.data_ptr(), float(scale_coef),
                                                 out.data_ptr(), workspace.data_ptr(),
                                                 workspace.numel() * workspace.element_size(), st.data_ptr(), int(flags),
                                                 _stream())
        _lib.check(rc)
        if _debug:
            check_status(tab.device)
        return out
    with torch.cuda.device(tab.device):
        rc = lib.sympa_all_pairs_dist(tab.data_ptr(), num_rows, n, int(row_begin), row_count, MODEL_IDS[model],
                                      METRIC_IDS[metric], None if w is None else w.data_ptr(), eps,
                                      None if sc is None else sc.data_ptr(), float(scale_coef), out.data_ptr(),
                                      st.data_ptr(), int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(tab.device)
    return out


# ---------------------------------------------------------------------------------------------------
# SPD model (C-ABI sympa_spd_dist_fwd / sympa_spd_model_forward)
# ---------------------------------------------------------------------------------------------------
def spd_dist_forward(x, y, flags=0):
    lib = _lib.load()
    _need_gpu(x, "x"); _need_gpu(y, "y")
    if x.dtype != torch.float64 or x.shape != y.shape or x.dim() != 3 or x.shape[1] != x.shape[2]:
        raise ValueError(f"expected two float64 [b,n,n] tensors, got {tuple(x.shape)} and {tuple(y.shape)}")
    x, y = x.detach().contiguous(), y.detach().contiguous()
    _gate(_sc.SPD_FWD, "spd", x.shape[1], x.device)
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    st = _status_buf(x.device)
    with torch.cuda.device(x.device):
        rc = lib.sympa_spd_dist_fwd(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), st.data_ptr(),
                                    int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(x.device)
    return out


def spd_model_forward(table, triplets, scale=None, scale_coef=1.0, out=None, flags=0):
    lib = _lib.load()
    _need_gpu(table, "table"); _need_gpu(triplets, "triplets")
    if table.dtype != torch.float64 or table.dim() != 3 or table.shape[1] != table.shape[2]:
        raise ValueError(f"spd table must be float64 [N,n,n], got {tuple(table.shape)}")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
        raise TypeError("triplets must be an int64 [b, >=2] tensor")
    tab = table.detach()
    tab = tab if tab.is_contiguous() else tab.contiguous()
    b = triplets.shape[0]
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    if out is None:
        out = torch.empty(b, dtype=torch.float64, device=tab.device)
    if b == 0:
        return out
    _gate(_sc.SPD_FWD, "spd", tab.shape[1], tab.device)
    sc = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=tab.device, dtype=torch.float64).contiguous()
    st = _status_buf(tab.device)
    tp = triplets.data_ptr()
    with torch.cuda.device(tab.device):
        rc = lib.sympa_spd_model_forward(tab.data_ptr(), tab.shape[0], tab.shape[1], tp, stride, tp + 8, stride, b,
                                         None if sc is None else sc.data_ptr(), float(scale_coef), out.data_ptr(),
                                         st.data_ptr(), int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(tab.device)
    return out


class SpdPackedTable(PackedTable):
    """PackedTable of the spd model (C-ABI sympa_spd_table_pack, n = 6..16): per point the n x n image [point's upper triangle |
    unit factor Lh of x = Lh D Lh^T in the strict lower triangle] + D^-1/2, made once per table state -- the pair kernel then
    skips the factorisation.  Same life cycle and device-side validity as PackedTable (`current` / `ensure` / `invalidate`).
    Not offered (`supported` is False, Model.forward keeps the dense entry) while the self-check has demoted the sixteen-lanes
    spd forward of these dims (selfcheck.py; the C entry itself then runs the one-lane kernel over the pack)."""

    @staticmethod
    def supported(table, model="spd"):
        if not (table.is_cuda and table.dtype == torch.float64 and table.dim() == 3 and table.shape[1] == table.shape[2]
                and table.shape[1] in SPD_PACKED_DIMS and table.is_contiguous()):
            return False
        return not _lib.load().sympa_get_instance_fallback(_sc.SPD_FWD, 0, int(table.shape[1]))

    def __init__(self, model="spd"):
        super().__init__("spd")

    @staticmethod
    def _key_of(table):
        return (table.untyped_storage().data_ptr(), table.storage_offset(), table.shape[0], table.shape[1], table._version,
                table.device)

    def _need_bytes(self, lib, table):
        return int(lib.sympa_spd_table_pack_bytes(table.shape[0], table.shape[1]))

    def _refresh(self, lib, table, need, force, stream):
        return lib.sympa_spd_table_pack_refresh(table.data_ptr(), table.shape[0], table.shape[1], self.pack.data_ptr(), need,
                                                self.state.data_ptr(), 1 if force else 0, None, stream)

    def _dims_of(self, table):
        return table.shape[1]


# the packed spd forward is used where it measured faster (profiles/r05_spd_packed_forward.txt: n = 13..15 spill at 256 registers
# and lose 6-32 %; n = 6, 8 gain less than a pack costs); the C-ABI serves every n = 6..16
SPD_PACKED_DIMS = frozenset({9, 10, 11, 12, 16})


def spd_model_forward_packed(packed, triplets, scale=None, scale_coef=1.0, out=None):
    """spd Model.forward over an SpdPackedTable (`packed.ensure(table)` first): C-ABI sympa_spd_model_forward_packed."""
    lib = _lib.load()
    _need_gpu(triplets, "triplets")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
        raise TypeError("triplets must be an int64 [b, >=2] tensor")
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    dev = packed.pack.device
    b = triplets.shape[0]
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    if out is None:
        out = torch.empty(b, dtype=torch.float64, device=dev)
    if b == 0:
        return out
    _gate(_sc.SPD_FWD, "spd", packed.n, dev)
    sc = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=dev, dtype=torch.float64).contiguous()
    tp = triplets.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.sympa_spd_model_forward_packed(packed.pack.data_ptr(), packed.bytes, packed.num_rows, packed.n, tp, stride, tp + 8,
                                                stride, b, None if sc is None else sc.data_ptr(), float(scale_coef),
                                                out.data_ptr(), _status_buf(dev).data_ptr(), 0, _stream())
    _lib.check(rc)
    if _debug:
        check_status(dev)
    return out


def _spd_bwd_workspace(lib, b, n, dev, flags, workspace):
    """The caller-owned scratch of the three-phase spd backward (C-ABI sympa_spd_backward_workspace_bytes; 0 bytes where no kernel
    uses one): `workspace` when given (a persistent uint8 tensor: what a replayed graph wants), else a fresh tensor -- the caching
    allocator serves it stream-ordered (inside a hipGraph capture from the graph's pool).  SYMPA_SPD_BWD_NO_WORKSPACE=1 (A/B) and
    FLAG_COOP / FLAG_GENERIC keep the kernels that need none."""
    if (flags & (FLAG_COOP | FLAG_GENERIC)) or os.environ.get("SYMPA_SPD_BWD_NO_WORKSPACE"):
        return None, 0
    # four launches instead of one: below ~1 000 pairs the launch latencies outweigh the saved QL (measured equal at 4 099);
    # a caller that hands over its own workspace gets the three-kernel path whatever the batch
    if workspace is None and b < int(os.environ.get("SYMPA_SPD_BWD_WORKSPACE_MIN", "1024")):
        return None, 0
    need = int(lib.sympa_spd_backward_workspace_bytes(int(b), int(n)))
    if need <= 0:
        return None, 0
    if workspace is not None:
        if workspace.dtype != torch.uint8 or not workspace.is_cuda or workspace.numel() < need or workspace.data_ptr() % 16:
            raise ValueError(f"workspace: a 16-byte aligned uint8 device tensor of at least {need} bytes")
        return workspace, workspace.numel()
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws, need


def spd_backward_rows(x, y, triplets=None, grad_out=None, graph_dist=None, scale=None, scale_coef=1.0, loss_scale=1.0,
                      loss=None, grad_scale=None, rows=None, want_out=False, flags=0, workspace=None):
    """Backward of the spd distance (C-ABI sympa_spd_backward_rows).  triplets None: pair i = (x[i], y[i]); otherwise
    x is y is the [N, n, n] table and pair i = (table[triplets[i, 0]], table[triplets[i, 1]]).  Give grad_out [b] or
    graph_dist [b] (fused AverageDistortionLoss, accumulated into `loss`).  Returns (rows [2b, n, n], out or None):
    rows [0, b) = gradient rows of the first points, [b, 2b) of the second."""
    lib = _lib.load()
    _need_gpu(x, "x"); _need_gpu(y, "y")
    if x.dtype != torch.float64 or x.dim() != 3 or x.shape[1] != x.shape[2]:
        raise ValueError(f"spd points / table must be float64 [., n, n], got {tuple(x.shape)}")
    x = x.detach(); y = y.detach()
    x = x if x.is_contiguous() else x.contiguous()
    y = y if y.is_contiguous() else y.contiguous()
    n = x.shape[1]
    dev = x.device
    _gate(_sc.SPD_BWD, "spd", n, dev)
    if triplets is not None:
        if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
            raise TypeError("triplets must be an int64 [b, >=2] tensor")
        if triplets.stride(1) != 1:
            triplets = triplets.contiguous()
        b = triplets.shape[0]
        stride = triplets.stride(0) if b > 1 else triplets.shape[1]
        sp, dp = triplets.data_ptr(), triplets.data_ptr() + 8
    else:
        b, stride, sp, dp = x.shape[0], 0, None, None
    if rows is None:
        rows = torch.empty(2 * b, n, n, dtype=torch.float64, device=dev)
    elif rows.numel() < 2 * b * n * n or not rows.is_contiguous():
        raise ValueError("rows buffer too small")
    out = torch.empty(b, dtype=torch.float64, device=dev) if want_out else None
    if b == 0:
        return rows, out
    go = None if grad_out is None else grad_out.detach().to(torch.float64).contiguous()
    gd = None if graph_dist is None else graph_dist.detach().to(torch.float64).contiguous()
    sc = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=dev, dtype=torch.float64).contiguous()
    st = _status_buf(dev)
    ws, ws_bytes = _spd_bwd_workspace(lib, b, n, dev, flags, workspace)
    with torch.cuda.device(dev):
        rc = lib.sympa_spd_backward_rows(
            x.data_ptr(), y.data_ptr(), x.shape[0], n, sp, stride, dp, stride, b, None if sc is None else sc.data_ptr(),
            float(scale_coef), None if go is None else go.data_ptr(), None if gd is None else gd.data_ptr(),
            float(loss_scale), None if loss is None else loss.data_ptr(), rows.data_ptr(),
            rows.data_ptr() + b * n * n * 8, None if grad_scale is None else grad_scale.data_ptr(),
            None if out is None else out.data_ptr(), st.data_ptr(), None if ws is None else ws.data_ptr(), int(ws_bytes),
            int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(dev)
    return rows, out


def spd_loss_backward(table, triplets, grad_table, grad_out=None, graph_dist=None, scale=None, scale_coef=1.0,
                      loss_scale=1.0, loss=None, grad_scale=None, want_out=False, flags=0, workspace=None):
    """spd backward with the scatter inside the kernel (C-ABI sympa_spd_loss_backward, n >= 3): the gradient rows of pair
    i = (table[triplets[i, 0]], table[triplets[i, 1]]) are ACCUMULATED into grad_table [N, n, n]; grad_out [b] or
    graph_dist [b] (fused AverageDistortionLoss into `loss`) as in spd_backward_rows.  Returns out [b] or None."""
    lib = _lib.load()
    _need_gpu(table, "table"); _need_gpu(triplets, "triplets"); _need_gpu(grad_table, "grad_table")
    if table.dtype != torch.float64 or table.dim() != 3 or table.shape[1] != table.shape[2]:
        raise ValueError(f"spd table must be float64 [N, n, n], got {tuple(table.shape)}")
    if grad_table.dtype != torch.float64 or not grad_table.is_contiguous() or grad_table.shape != table.shape:
        raise ValueError("grad_table must be a contiguous float64 tensor of the table's shape")
    if triplets.dtype != torch.int64 or triplets.dim() != 2 or triplets.shape[1] < 2:
        raise TypeError("triplets must be an int64 [b, >=2] tensor")
    tab = table.detach()
    tab = tab if tab.is_contiguous() else tab.contiguous()
    if triplets.stride(1) != 1:
        triplets = triplets.contiguous()
    b, n, dev = triplets.shape[0], tab.shape[1], tab.device
    out = torch.empty(b, dtype=torch.float64, device=dev) if want_out else None
    if b == 0:
        return out
    _gate(_sc.SPD_BWD, "spd", n, dev)
    if n < 3 or (flags & FLAG_GENERIC) or lib.sympa_get_instance_fallback(_sc.SPD_BWD, 0, n):
        # no one-lane kernel has the scatter inside: per-pair rows, then the coalesced scatter-add (two launches)
        rows, out = spd_backward_rows(tab, tab, triplets, grad_out=grad_out, graph_dist=graph_dist, scale=scale,
                                      scale_coef=scale_coef, loss_scale=loss_scale, loss=loss, grad_scale=grad_scale,
                                      want_out=want_out, flags=flags)
        idx = torch.cat((triplets[:, 0], triplets[:, 1])).contiguous()
        scatter_add_flat_rows_(grad_table, rows[:2 * b].reshape(2 * b, -1), idx)
        return out
    stride = triplets.stride(0) if b > 1 else triplets.shape[1]
    go = None if grad_out is None else grad_out.detach().to(torch.float64).contiguous()
    gd = None if graph_dist is None else graph_dist.detach().to(torch.float64).contiguous()
    sc = None
    if scale is not None:
        sc = scale.detach().reshape(-1)[:1].to(device=dev, dtype=torch.float64).contiguous()
    st = _status_buf(dev)
    tp = triplets.data_ptr()
    ws, ws_bytes = _spd_bwd_workspace(lib, b, n, dev, flags, workspace)
    with torch.cuda.device(dev):
        rc = lib.sympa_spd_loss_backward(
            tab.data_ptr(), tab.shape[0], n, tp, stride, tp + 8, stride, b, None if sc is None else sc.data_ptr(),
            float(scale_coef), None if go is None else go.data_ptr(), None if gd is None else gd.data_ptr(),
            float(loss_scale), None if loss is None else loss.data_ptr(), grad_table.data_ptr(),
            None if grad_scale is None else grad_scale.data_ptr(), None if out is None else out.data_ptr(),
            st.data_ptr(), None if ws is None else ws.data_ptr(), int(ws_bytes), int(flags), _stream())
    _lib.check(rc)
    if _debug:
        check_status(dev)
    return out


def scatter_add_flat_rows_(grad_table, rows, idx, alpha=1.0):
    """grad_table[idx[r]] += alpha * rows[r] for rows of any length (C-ABI sympa_scatter_add_flat_rows)."""
    lib = _lib.load()
    _need_gpu(grad_table, "grad_table"); _need_gpu(rows, "rows"); _need_gpu(idx, "idx")
    if grad_table.dtype != torch.float64 or rows.dtype != torch.float64 or idx.dtype != torch.int64:
        raise TypeError("float64 rows / table and int64 indices expected")
    if not (grad_table.is_contiguous() and rows.is_contiguous() and idx.dim() == 1):
        raise ValueError("contiguous tensors and a 1-d index list expected")
    count = idx.shape[0]
    rowd = grad_table[0].numel()
    if rows.numel() < count * rowd:
        raise ValueError("rows must hold one gradient row per index")
    st = _status_buf(grad_table.device)
    with torch.cuda.device(grad_table.device):
        rc = lib.sympa_scatter_add_flat_rows(rows.data_ptr(), idx.data_ptr(), idx.stride(0), count, rowd,
                                             grad_table.shape[0], float(alpha), grad_table.data_ptr(), st.data_ptr(), _stream())
    _lib.check(rc)
    return grad_table


def _spd_rows(t, name):
    _need_gpu(t, name)
    if t.dtype != torch.float64 or t.dim() != 3 or t.shape[1] != t.shape[2]:
        raise ValueError(f"{name} must be a float64 [b, n, n] tensor, got {tuple(t.shape)} {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def spd_egrad2rgrad(x, u):
    lib = _lib.load()
    x, u = _spd_rows(x.detach(), "x"), _spd_rows(u.detach(), "u")
    _gate(_sc.SPD_TABLE, "spd", x.shape[1], x.device)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.sympa_spd_egrad2rgrad(x.data_ptr(), u.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream())
    _lib.check(rc)
    return out


def spd_projx(x, counter=None):
    lib = _lib.load()
    x = _spd_rows(x.detach(), "x")
    out = torch.empty_like(x)
    st = _status_buf(x.device)
    with torch.cuda.device(x.device):
        rc = lib.sympa_spd_projx(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(),
                                 None if counter is None else counter.data_ptr(), st.data_ptr(), _stream())
    _lib.check(rc)
    return out


def spd_rsgd_step_(table, grad, lr, weight_decay=0.0, clip_sqnorm=None, max_norm=None):
    """In-place geoopt RiemannianSGD step over an spd table (C-ABI sympa_spd_rsgd_step)."""
    lib = _lib.load()
    _need_gpu(table, "table")
    if not table.is_contiguous():
        raise ValueError("table must be contiguous for the in-place step")
    grad = _spd_rows(grad.detach(), "grad")
    if table.dtype != torch.float64 or table.dim() != 3 or table.shape != grad.shape:
        raise ValueError(f"table must be a float64 [N, n, n] tensor of the gradient's shape {tuple(grad.shape)}, got "
                         f"{tuple(table.shape)} {table.dtype}")
    _gate(_sc.SPD_TABLE, "spd", table.shape[1], table.device)
    st = _status_buf(table.device)
    with torch.cuda.device(table.device):
        rc = lib.sympa_spd_rsgd_step(table.data_ptr(), grad.data_ptr(), table.shape[0], table.shape[1], float(lr),
                                     float(weight_decay), None if clip_sqnorm is None else clip_sqnorm.data_ptr(),
                                     float(max_norm) if max_norm is not None else 0.0, st.data_ptr(), _stream())
    _lib.check(rc)
    return table
