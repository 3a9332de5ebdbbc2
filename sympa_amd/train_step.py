"""The body of the reference's training loop (sympa/runner.py:98-118: zero_grad, forward, AverageDistortionLoss,
backward, gradient clip, optimiser step) as ONE hipGraph replay per batch.

Eagerly the step is a handful of launches issued from Python; at the reference's batch sizes (512 .. 8192 triplets)
the Python and launch overhead (~0.3 ms) is several times the GPU time.  Everything in the step is stream-ordered and
allocation-free after the first call, so it is captured once per (batch size, learning rate) and replayed.

Two graph shapes:

* **two kernels** (Siegel models, dims <= 6, RiemannianSGD, table of at most CUs x 256 rows): `sympa_model_train_backward`
  (distances + loss + backward + scatter) and `sympa_rsgd_step_fused` (clip norm, table step, scale / wsum-weight step,
  zero_grad).  The batch is addressed through a device step counter that the second kernel increments, so an epoch's
  shuffled triplets are loaded ONCE (`load_epoch`) and `run_steps(k)` is k replays with no copy, no memset and no host
  arithmetic in between.  `deterministic=True` swaps the atomic scatter for per-pair gradient rows + a segmented sum in
  a precomputed order (one sort per epoch) and the scalar atomics for fixed-order sums: two runs of an epoch give the
  same bits, like the reference's CPU autograd.  From 4 096 pairs per batch the batches are sorted by source inside
  `load_epoch` (the order inside a batch is free) and the backward kernel writes every run of equal source ids of a wave as
  ONE row (`merges_source_rows`, SYMPA_FLAG_MERGE_SRC): 45 % fewer rows through memory at the headline shape.
* **classic** (everything else: dims 7..16, spd, other optimisers' parameters): zero (one multi-tensor launch), the fused
  loss + backward kernel, squared norms, the RSGD kernel(s), the scale's step -- round 2's graph."""
import os

import torch

from sympa_amd import _lib, data, ops
from sympa_amd import selfcheck as _sc
from sympa_amd.manifolds.metrics import MetricType


def merges_source_rows(model, deterministic, batch_size=None):
    """True where the training backward of the Siegel models at dims <= 6 merges runs of equal source ids inside a wave (round 5:
    SYMPA_FLAG_MERGE_SRC, csrc/siegel_bwd_kernel.hpp) -- the deterministic rows form writes ONE row per run (store_rows_merged: 45 %
    fewer gradient rows written and read back at the headline shape), the atomic form sums the run in its LDS tile and adds one row
    (scatter_add_rows) -- on batches sorted by source (load_epoch).  Measured, per step: deterministic 55.3 -> 49.9 us at the headline
    shape, configs[2] 60.7 -> 55.4, configs[1] (8 192 pairs) 38.8 -> 38.2, configs[0] (512 pairs) 21.7 -> 23.8: from 4 096 pairs per
    batch; atomic 63.1 -> 58.4 / 66.7 -> 63.6, but 32.9 -> 35.1 at 8 192 pairs: from 32 768.  SYMPA_MERGE_SRC_MIN=<pairs> moves both
    thresholds, SYMPA_NO_MERGE_SRC=1 switches the merging off."""
    if os.environ.get("SYMPA_NO_MERGE_SRC"):
        return False
    floor = int(os.environ.get("SYMPA_MERGE_SRC_MIN", "4096" if deterministic else "32768"))
    if batch_size is not None and int(batch_size) < floor:
        return False
    table = model.embeddings.embeds
    return getattr(model.manifold, "model_name", "") in ("upper", "bounded") and table.dim() == 4 and int(table.shape[-1]) <= 6


def batches_want_source_order(model, batch_size=None, deterministic=False):
    """True where the backward's scatter merges consecutive pairs with the same source row -- the split backward of the upper
    model at dims 8 (csrc/siegel_bwd_split_kernel.hpp) and the three-kernel spd backward at dims 9..16 (spd_coop_bwd3_kernel.hpp):
    there a batch sorted by its first column (data.sort_batches_by_source) is 10-15 % faster.  Everywhere else the sorted order puts
    the atomics of neighbouring lanes on the SAME rows and is slower (headline two-kernel step 62 -> 71 us, dims 7 fused step 200 ->
    228 us per 65 536 pairs), so the batches are left in the sampler's order.  The merging kernels run from 1 024 pairs per batch
    on (ops: SYMPA_SIEGEL_BWD_WORKSPACE_MIN / SYMPA_SPD_BWD_WORKSPACE_MIN; smaller batches stay with the one-launch kernels, where
    a sorted batch is slower), so with `batch_size` given the answer is False below that."""
    man = model.manifold
    dims = int(model.embeddings.embeds.shape[-1])
    name = getattr(man, "model_name", "")
    if os.environ.get("SYMPA_NO_BATCH_SORT"):
        return False
    if merges_source_rows(model, deterministic, batch_size):          # the rows form adds no atomics: the sorted order only shortens the lists
        return True
    if name == "upper" and dims == 8:
        floor = int(os.environ.get("SYMPA_SIEGEL_BWD_WORKSPACE_MIN", "1024"))
    elif name == "spd" and 9 <= dims <= 16:
        floor = int(os.environ.get("SYMPA_SPD_BWD_WORKSPACE_MIN", "1024"))
    else:
        return False
    return batch_size is None or int(batch_size) >= floor


class GraphedTrainStep:
    def __init__(self, model, optimizer, batch_size, max_grad_norm, device, deterministic=False, accumulate_loss=False,
                 two_kernels=True):
        # the capture bakes every host-side scalar of the optimiser's step in as an immediate: only optimisers whose step has
        # no host state that changes from step to step may be captured (sympa_amd.optim.RiemannianSGD; RiemannianAdam keeps
        # its powers b^t in device words for this reason).  An optimiser with moments also provides init_state /
        # snapshot_state / restore_state, so that the warm-up steps of the capture do not count.
        if not getattr(optimizer, "graph_capturable", False):
            raise TypeError(f"{type(optimizer).__name__} cannot be captured in a hipGraph (its step() changes host-side state "
                            "every call); use sympa_amd.optim.RiemannianSGD / RiemannianAdam or run the step eagerly")
        self.model, self.opt = model, optimizer
        self.batch_size, self.max_grad_norm = int(batch_size), float(max_grad_norm)
        self.device = torch.device(device)
        self.accumulate_loss = bool(accumulate_loss)
        self.loss = torch.zeros(1, dtype=torch.float64, device=device)
        self.graph = None
        self.key = None
        self.grad_ptrs = []
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._zero_list = None
        self.mode = "two_kernels" if (two_kernels and self._two_kernels_possible()) else "classic"
        # deterministic=None: take the deterministic accumulation where it is also the faster form -- large batches, where
        # the atomic scatter costs more than per-pair rows + the segmented sum (headline shape, 65 536 pairs: 55 against
        # 62.5 us per step; at 8 192 pairs its third launch costs more than it saves: 41 against 32.5 us)
        if deterministic is None:
            deterministic = self.mode == "two_kernels" and self.batch_size >= 32768
        self.deterministic = bool(deterministic)
        if self.deterministic and self.mode != "two_kernels":
            raise ValueError("deterministic accumulation is built for the two-kernel step (Siegel models, dims <= 6, "
                             "RiemannianSGD, tables of at most CUs x 256 rows)")
        # the batches of an epoch (two-kernel mode) / the static batch (classic mode)
        self.capacity = self.batch_size
        self._alloc_epoch(self.batch_size)
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.steps_loaded = 0
        self._fused = None
        self._fused_sig = None
        self.state_ptrs = None

    # ------------------------------------------------------------------------------------------------------------
    def _two_kernels_possible(self):
        from sympa_amd.optim import RiemannianAdam, RiemannianSGD
        m = self.model
        man = m.manifold
        table = m.embeddings.embeds
        if not isinstance(self.opt, (RiemannianSGD, RiemannianAdam)) or getattr(man, "model_name", None) not in ("upper", "bounded"):
            return False
        if isinstance(self.opt, RiemannianAdam) and len({(tuple(g["betas"]), g["eps"]) for g in self.opt.param_groups}) != 1:
            return False            # one set of Adam hyper-parameters per launch
        if not ops.FusedStep.supported(table.data):
            return False
        extras = [p for p in self.params if p is not table]
        if not table.requires_grad or len(extras) > 2 or any(p.numel() > 64 or p.dtype != torch.float64 or not p.is_cuda
                                                            for p in extras):
            return False
        known = [m.scale] + ([man.metric.weights] if man.metric.kind is MetricType.WEIGHTED_SUM else [])
        if any(all(p is not k for k in known) for p in extras):
            return False
        in_opt = [p for g in self.opt.param_groups for p in g["params"]]
        return all(any(p is q for q in in_opt) for p in self.params)

    def _alloc_epoch(self, pairs):
        dev = self.device
        self.capacity = int(pairs)
        self.ids = torch.zeros(self.capacity, 2, dtype=torch.int64, device=dev)
        self.gd = torch.ones(self.capacity, dtype=torch.float64, device=dev)
        if self.deterministic:
            steps = max(1, self.capacity // self.batch_size)
            n_rows = self.model.embeddings.embeds.shape[0]
            self.order = torch.zeros(steps, 2 * self.batch_size, dtype=torch.int32, device=dev)
            self.rowptr = torch.zeros(steps, n_rows + 1, dtype=torch.int32, device=dev)
        self.graph = None            # the captured launches hold the old addresses

    def _merge_flags(self):
        if not self.__dict__.get("_sorted_by_source", True):       # (a fresh object assumes the load_epoch path)
            return 0
        return ops.FLAG_MERGE_SRC if merges_source_rows(self.model, self.deterministic, self.batch_size) else 0

    def _bwd_ws(self, table, b):
        """The persistent scratch of the split Siegel backward for batches of b pairs (None where no kernel uses one): a replayed
        graph must not allocate, so it is made once per batch size and kept."""
        cache = self.__dict__.setdefault("_bwd_ws_cache", {})
        key = (int(b), table.shape[2], table.device)
        if key not in cache:
            cache[key] = ops.siegel_backward_workspace(b, table.shape[2], self.model.manifold.model_name, table.device) \
                if table.dim() == 4 else None
        return cache[key]

    def _group_of(self, p):
        for g in self.opt.param_groups:
            if any(p is q for q in g["params"]):
                return g
        raise KeyError("parameter not in the optimiser")

    # ------------------------------------------------------------------------------------------------------------
    def _backward(self, counter):
        """distances + loss + gradients of batch `counter` of the loaded triplets (two-kernel mode)."""
        m = self.model
        man = m.manifold
        table = m.embeddings.embeds
        wsum = man.metric.kind is MetricType.WEIGHTED_SUM
        weights = man.metric.weights if wsum else None
        gw = weights.grad if wsum else None
        gs = m.scale.grad if m.scale.requires_grad else None
        b = self.batch_size
        if not self.deterministic:
            ops.model_train_backward(table.data, self.ids, self.gd, b, self.loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_table=table.grad, step_counter=counter, workspace=self._bwd_ws(table, b),
                                     flags=self._merge_flags())
            return
        ops.model_train_backward(table.data, self.ids, self.gd, b, self.loss, man.model_name, man.metric.kind.value,
                                 None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                 grad_rows=self.rows, step_counter=counter, wave_partials=self.partials, workspace=self._bwd_ws(table, b),
                                 flags=self._merge_flags())
        ops.segment_sum_rows_(table.grad, self.rows, self.order, self.rowptr, step_counter=counter,
                              wave_partials=self.partials, num_waves=(b + 63) // 64, partial_stride=2 + table.shape[2],
                              loss=self.loss, grad_scale=gs, grad_weights=gw, sq_partials=self.sq_partials)

    def _body(self):
        if self.mode == "two_kernels":
            if not self.accumulate_loss:
                self.loss.zero_()
            self._backward(self.counter)
            self._fused_step()
            return
        # classic graph.  Siegel models (round 6): the backward addresses its batch through the device step counter like the
        # two-kernel step (C-ABI sympa_model_train_backward), so an epoch's triplets are loaded ONCE (load_epoch / run_steps) and no
        # batch is copied in per step (configs[3]: two copies of 4 + 2 MB, ~10 us of the 725 us step); __call__ zeroes the counter and
        # copies its batch to the front of the same buffers.  The counter moves at the end of the graph.
        windowed = self._classic_windowed()
        if self._zero_list is None:
            self.opt.zero_grad(set_to_none=False)
            if windowed:
                if not self.accumulate_loss:
                    self.loss.zero_()
                self._backward(self.counter)
            else:
                self.model.fused_loss_backward(self.ids[:self.batch_size], self.gd[:self.batch_size],
                                               loss_out=self.loss, zero_loss_out=not self.accumulate_loss)
            self._clip_and_step()
            if windowed:
                self.counter.add_(1)
            return
        # gradients, the loss word and the optimiser's squared-norm word zeroed by ONE multi-tensor launch
        torch._foreach_zero_(self._zero_list)
        if windowed:
            self._backward(self.counter)
        else:
            self.model.fused_loss_backward(self.ids[:self.batch_size], self.gd[:self.batch_size], loss_out=self.loss,
                                           zero_loss_out=False)
        self.opt._sqnorm_zeroed_by_caller = True
        try:
            self._clip_and_step()
        finally:
            self.opt._sqnorm_zeroed_by_caller = False
        if windowed:
            self.counter.add_(1)

    def _classic_windowed(self):
        """Classic graph of a Siegel model on the GPU with dims <= 8 or the sixteen-lanes kernels (every family that honours the
        batch window; the rolled one-lane kernels of dims 9..16 behind an instance fallback do not)."""
        m = self.model
        if self.mode != "classic" or getattr(m.manifold, "model_name", None) not in ("upper", "bounded") or self.deterministic:
            return False
        table = m.embeddings.embeds
        if not (table.is_cuda and table.dim() == 4 and table.dtype == torch.float64):
            return False
        n = table.shape[2]
        if n > 8 and _lib.load().sympa_get_instance_fallback(_sc.SIEGEL_BWD, ops.MODEL_IDS[m.manifold.model_name], int(n)):
            return False
        return True

    def _clip_and_step(self):
        if hasattr(self.opt, "clip_max_norm"):       # sympa_amd.optim.RiemannianSGD: the clip rides inside the step
            self.opt.clip_max_norm = self.max_grad_norm
            self.opt.step()
            self.opt.clip_max_norm = None
        else:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
            self.opt.step()

    def _ensure_fused(self):
        """Gradient tensors (zero) and the fused optimiser kernel's plan (two-kernel mode)."""
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p.data)
        if self.mode != "two_kernels":
            return
        if self._fused is not None and self._fused_sig == self._fused_signature():
            return
        table = self.model.embeddings.embeds
        man = self.model.manifold
        self._extra_params = [p for p in self.params if p is not table]
        extras = [(p.data, p.grad) for p in self._extra_params]
        # zero_grads: the scatter accumulates into the table gradient, the scalar sums into the scale / weight gradients
        self.sq_partials = None
        if self.deterministic:      # the segmented sum leaves the squared-norm partials: the optimiser kernel needs no barrier
            self.sq_partials = torch.zeros(ops.segment_sum_partials(table.grad), dtype=torch.float64, device=self.device)
        adam = None
        if hasattr(self.opt, "snapshot_state"):         # RiemannianAdam: its state tensors go into the kernel's plan
            self.opt.init_state()
            st = self.opt.state
            g0 = self.opt.param_groups[0]
            adam = dict(exp_avg=st[table]["exp_avg"], exp_avg_sq=st[table]["exp_avg_sq"], bias_pows=st[table]["bias_pows"],
                        betas=g0["betas"], eps=g0["eps"],
                        extras=[(st[p]["exp_avg"], st[p]["exp_avg_sq"], st[p]["bias_pows"]) for p in self._extra_params])
        self._fused = ops.FusedStep(table.data, table.grad, man.model_name, extras, counter=self.counter,
                                    projected=man.projected_counter(table.device), zero_grads=True,
                                    sq_partials=self.sq_partials, adam=adam)
        self._fused_sig = self._fused_signature()
        if self.deterministic:
            n = table.shape[2]
            self.rows = torch.empty(2 * self.batch_size, 2, n, n, dtype=torch.float64, device=self.device)
            self.partials = torch.zeros((self.batch_size + 63) // 64, 2 + n, dtype=torch.float64, device=self.device)
        for p in self.params:          # the step leaves the gradients zero; it must find them zero the first time
            p.grad.zero_()

    def _fused_step(self):
        tg = self._group_of(self.model.embeddings.embeds)
        xg = [self._group_of(p) for p in self._extra_params]
        self._fused.run(tg["lr"], tg.get("weight_decay", 0.0), self.max_grad_norm,
                        [g["lr"] for g in xg], [g.get("weight_decay", 0.0) for g in xg])

    def eager(self, ids, gd):
        """The same step without a graph (ragged last batch of an epoch, multi-GPU steps with an all-reduce inside).
        Two-kernel mode: the same two kernels launched directly (deterministic: per-pair rows + one sort + the segmented
        sum, so the epoch stays reproducible).  Returns the batch loss (a fresh 1-element tensor)."""
        if self.mode != "two_kernels":
            self.opt.zero_grad(set_to_none=False)
            loss = self.model.fused_loss_backward(ids, gd)
            self._clip_and_step()
            return loss
        self._ensure_fused()
        m = self.model
        man = m.manifold
        table = m.embeddings.embeds
        b = ids.shape[0]
        wsum = man.metric.kind is MetricType.WEIGHTED_SUM
        weights = man.metric.weights if wsum else None
        gw = weights.grad if wsum else None
        gs = m.scale.grad if m.scale.requires_grad else None
        n = table.shape[2]
        loss = torch.zeros(1, dtype=torch.float64, device=table.device)
        ids = ids[:, :2].contiguous()
        gd = gd.to(torch.float64).contiguous()
        if b > 0 and self._merge_flags():           # (as in __call__: runs of equal source ids exist in sorted batches only)
            order = torch.argsort(ids[:, 0], stable=True)
            ids, gd = ids[order].contiguous(), gd[order].contiguous()
        if b > 0 and not self.deterministic:
            ops.model_train_backward(table.data, ids, gd, b, loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_table=table.grad, workspace=self._bwd_ws(table, b), flags=self._merge_flags())
        elif b > 0:
            rows = torch.empty(2 * b, 2, n, n, dtype=torch.float64, device=table.device)
            partials = torch.empty((b + 63) // 64, 2 + n, dtype=torch.float64, device=table.device)
            ops.model_train_backward(table.data, ids, gd, b, loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_rows=rows, wave_partials=partials, workspace=self._bwd_ws(table, b), flags=self._merge_flags())
            order, rowptr = ops.sorted_slots(torch.cat((ids[:, 0], ids[:, 1])), table.shape[0], merged_src=b if self._merge_flags() else 0)
            ops.segment_sum_rows_(table.grad, rows, order, rowptr, wave_partials=partials, num_waves=(b + 63) // 64,
                                  partial_stride=2 + n, loss=loss, grad_scale=gs, grad_weights=gw,
                                  sq_partials=self.sq_partials)
        if b == 0:
            return loss                 # nothing to step on (and the deterministic clip would read the last step's partials)
        self._fused_step()
        ops.table_changed(table)
        # the fused optimiser kernel advances the device step counter that addresses the batches of a loaded epoch; an
        # eager step is not one of them, so the window stays where run_steps left it (stream-ordered, no sync)
        self.counter.sub_(1)
        return loss

    def _key(self):
        """Everything the captured launches bake in as immediates: per-group lr and weight decay, the clip norm, and (Adam)
        betas / eps."""
        return (tuple((float(g["lr"]), float(g.get("weight_decay", 0.0)),
                       tuple(float(x) for x in g["betas"]) if "betas" in g else None,
                       float(g["eps"]) if "eps" in g else None) for g in self.opt.param_groups),
                self.max_grad_norm, self._merge_flags())

    def _fused_signature(self):
        """What the fused optimiser kernel's plan bakes in beside lr / weight decay (passed per run): the addresses of the
        optimiser's state tensors and Adam's betas / eps."""
        return (self._state_ptrs(), tuple((tuple(float(x) for x in g["betas"]), float(g["eps"]))
                                          for g in self.opt.param_groups if "betas" in g))

    def _state_ptrs(self):
        """Addresses of every optimiser state tensor the captured launches / the fused kernel's plan hold (RiemannianAdam's
        moments and power words): `opt.load_state_dict` replaces those tensors, and a replay would go on updating the
        orphaned ones."""
        out = []
        for p in self.params:
            st = self.opt.state.get(p, {})
            out.append(tuple((k, v.data_ptr()) for k, v in sorted(st.items()) if torch.is_tensor(v)))
        return out

    def _capture(self):
        if ops._debug:
            raise RuntimeError("ops.set_debug(True) synchronises after every kernel and cannot run inside a hipGraph "
                               "capture: switch it off for graphed training steps")
        self._ensure_fused()
        # warm-up outside the capture with lr = 0: allocates the status words / workspaces and leaves the parameters where
        # they are (retr(x, 0) = projx(x), the identity for points on the manifold); every group gets its own lr back
        saved = [g["lr"] for g in self.opt.param_groups]
        snap = None
        if hasattr(self.opt, "snapshot_state"):         # moments and step counts: the warm-up steps must leave no trace
            self.opt.init_state()
            snap = self.opt.snapshot_state()
        saved_loss = self.loss.clone()
        saved_counter = self.counter.clone()
        for g in self.opt.param_groups:
            g["lr"] = 0.0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.counter.zero_()
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        for g, lr in zip(self.opt.param_groups, saved):
            g["lr"] = lr
        if snap is not None:
            self.opt.restore_state(snap)
        self.loss.copy_(saved_loss)
        self.counter.copy_(saved_counter)
        # after the warm-up every buffer exists: from here on one foreach launch zeroes them all (classic mode)
        self._zero_list = None
        if self.mode == "classic" and hasattr(self.opt, "_sqnorm_zeroed_by_caller") and hasattr(self.opt, "clip_max_norm"):
            grads = [p.grad for p in self.params if p.grad is not None]
            words = ([] if self.accumulate_loss else [self.loss]) + \
                [t for t in getattr(self.opt, "_sqnorm", {}).values() if t.device == self.loss.device]
            if grads and all(g.is_cuda for g in grads):
                self._zero_list = grads + words
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._body()
        self.key = self._key()
        # the graph writes the gradients through these addresses
        self.grad_ptrs = [None if p.grad is None else p.grad.data_ptr() for p in self.params]
        self.state_ptrs = self._state_ptrs()

    def _ready(self):
        if self.graph is not None and self._state_ptrs() != self.state_ptrs:
            # the optimiser's state tensors were replaced (load_state_dict): the graph holds the old addresses (and so
            # does the fused kernel's plan, which _ensure_fused rebuilds on the same signature)
            self.graph = None
        if self.graph is None or self._key() != self.key:
            self._capture()          # learning rate / weight decay / clip norm / betas of any group changed (end of burn-in)
        for p, ptr in zip(self.params, self.grad_ptrs):
            if (None if p.grad is None else p.grad.data_ptr()) != ptr:
                # e.g. zero_grad(set_to_none=True) between replays: the graph would write freed memory
                raise RuntimeError("a parameter's .grad was replaced since the step was captured; keep the gradient "
                                   "tensors (zero_grad(set_to_none=False)) or build a new GraphedTrainStep")

    # ------------------------------------------------------------------------------------------------------------
    def __call__(self, ids, gd):
        """ids [b, 2] int64, gd [b] fp64 on the device.  Returns the loss word: the batch loss (accumulate_loss=False: it is
        overwritten by the next call -- add it to an accumulator, do not keep it) or the running sum since reset_loss()."""
        if ids.shape[0] != self.batch_size:
            loss = self.eager(ids, gd)
            if self.accumulate_loss:
                self.loss += loss
                return self.loss
            return loss
        b = self.batch_size
        # the merged-rows backward sums RUNS of equal source ids inside a wave: a caller-ordered batch has next to none and would only
        # pay the tile walk (round-5 advice).  Sorting it here costs more than the walk (argsort + two gathers: +67 us per step at the
        # headline shape, measured), so the flag is simply off for batches this object did not sort itself (load_epoch does);
        # the captured graph bakes the flag in and is re-captured when a caller switches between the two ways of feeding it.
        self._sorted_by_source = False
        self.ids[:b].copy_(ids[:, :2])
        self.gd[:b].copy_(gd)
        if self.mode == "classic" and self._classic_windowed():
            self.counter.zero_()
        if self.mode == "two_kernels":
            self.counter.zero_()
            if self.deterministic:
                order, rowptr = ops.sorted_slots(torch.cat((self.ids[:b, 0], self.ids[:b, 1])),
                                                 self.model.embeddings.embeds.shape[0], merged_src=b if self._merge_flags() else 0)
                self.order[0].copy_(order[0])
                self.rowptr[0].copy_(rowptr[0])
        self._ready()
        self.graph.replay()
        ops.table_changed(self.model.embeddings.embeds)        # a replayed graph has no Python in it: tell torch the table moved
        return self.loss

    def load_epoch(self, triplets):
        """triplets [T, 3] int64 (src, dst, graph distance) in the order the epoch visits them (the DistributedSampler
        shard, train.py:105-110).  Copies them into the step's persistent buffers once, builds the deterministic mode's
        sorted slot lists with ONE sort, resets the device step counter.  Returns the number of FULL batches
        `run_steps` may replay; the ragged remainder triplets[steps * batch:] goes through `eager` (or `__call__`)."""
        if self.mode != "two_kernels" and not self._classic_windowed():
            raise RuntimeError("load_epoch / run_steps need a step whose backward addresses its batch through the device step "
                               "counter (the Siegel models); this model runs the classic graph on a static batch: call the "
                               "object once per batch")
        total = triplets.shape[0]
        b = self.batch_size
        steps = total // b
        if total > self.capacity:
            self._alloc_epoch(total)
        if batches_want_source_order(self.model, b, self.deterministic):      # the order inside a batch is free: equal source rows adjacent
            triplets = data.sort_batches_by_source(triplets, b)
        self.ids[:total].copy_(triplets[:, :2])
        self.gd[:total].copy_(triplets[:, 2])
        self._sorted_by_source = True
        if self.deterministic and steps > 0:
            used = self.ids[:steps * b].view(steps, b, 2)
            order, rowptr = ops.sorted_slots(torch.cat((used[:, :, 0], used[:, :, 1]), dim=1),
                                             self.model.embeddings.embeds.shape[0], merged_src=b if self._merge_flags() else 0)
            self.order[:steps].copy_(order)
            self.rowptr[:steps].copy_(rowptr)
        self.counter.zero_()
        self.steps_loaded = steps
        return steps

    def run_steps(self, k=None):
        """k replays of the two-kernel graph on the next k batches of the loaded epoch (default: all that are left).
        Nothing else is enqueued between the replays."""
        self._ready()
        k = self.steps_loaded if k is None else int(k)
        if k > self.steps_loaded:
            raise ValueError("more steps than full batches left in the loaded epoch")
        for _ in range(k):
            self.graph.replay()
        if k > 0:
            ops.table_changed(self.model.embeddings.embeds)
        self.steps_loaded -= k
        return self.loss

    def reset_loss(self):
        self.loss.zero_()


class DistributedTrainStep:
    """The data-parallel training step (train.py:59,105-110,136 + runner.py:98-118: every rank runs backward on its shard of
    the global batch, the gradients are averaged, every rank takes the optimiser step) as REPLAYED GRAPHS around the
    exchange -- the multi-GPU counterpart of GraphedTrainStep's two-kernel step, RiemannianSGD, Siegel models.

        graph A   sympa_model_train_backward: distances + loss + backward of batch c (device step counter; an epoch's
                  shard is loaded once with `load_epoch`), scattered into the table-gradient view of GradientExchange's flat
                  buffer (mode "rows": per-pair rows + the batch's row indices gathered on the device)
        exchange  dense: ONE in-place all-reduce of the flat buffer on the same stream;  rows: all-gather of rows + indices,
                  scatter-add;  sharded: reduce-scatter / 8-byte all-reduce / all-gather around the shard's step
        graph B   clip + RiemannianSGD + scale / weight step + zero_grad + counter increment: sympa_rsgd_step_fused (one
                  launch) where the table qualifies, the separate kernels otherwise

    With a backend whose collectives are stream work (RCCL) the whole step is tried as ONE captured graph first
    (`capture_collective`; `graphs_per_step` says what was achieved: 1, or 2 with the exchange enqueued between the replays);
    the sharded mode's optimiser side runs between the graphs.  No host synchronisation anywhere in `run_steps`."""

    def __init__(self, model, optimizer, batch_size, max_grad_norm, device, mode="dense", group=None, capture_collective=None,
                 accumulate_loss=True, deterministic=None):
        import torch.distributed as dist
        from sympa_amd.distributed import GradientExchange
        from sympa_amd.optim import RiemannianSGD
        man = model.manifold
        table = model.embeddings.embeds
        if getattr(man, "model_name", None) not in ("upper", "bounded", "spd"):
            raise NotImplementedError("DistributedTrainStep: the Siegel models and spd")
        self.spd = man.model_name == "spd"
        if self.spd and mode == "auto":
            # the spd backward scatters a dense gradient and never writes per-pair rows: "auto" must not resolve to "rows"
            # (it would on a large graph at the reference's default batch sizes, and every step would apply a zero gradient)
            mode = "dense"
        if self.spd and mode == "rows":
            raise NotImplementedError("DistributedTrainStep: the spd model exchanges dense or sharded (configs[4]: 2 B > N)")
        if not isinstance(optimizer, RiemannianSGD):
            raise NotImplementedError("DistributedTrainStep: sympa_amd.optim.RiemannianSGD")
        self.model, self.opt, self.dist = model, optimizer, dist
        self.batch_size, self.max_grad_norm = int(batch_size), float(max_grad_norm)
        self.device = torch.device(device)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.ex = GradientExchange(self.params, table=table, local_batch=self.batch_size, mode=mode, group=group)
        self.mode = self.ex.mode
        if self.spd and self.mode == "rows":          # validated AFTER GradientExchange resolved the mode
            raise NotImplementedError("DistributedTrainStep: the spd model exchanges dense or sharded, not touched rows")
        self.world = self.ex.world
        self.accumulate_loss = bool(accumulate_loss)
        # deterministic local accumulation (dense / sharded modes, dims <= 6): per-pair gradient rows + the segmented sum in a
        # precomputed order + fixed-order scalar sums instead of fp64 atomics -- every rank's local gradient is bitwise
        # reproducible (the collective's own summation order is the backend's), and at large batches it is also the faster
        # form (GraphedTrainStep: 55 against 63 us at 65 536 triplets).  None: taken from 32 768 triplets per batch on.
        det_ok = self.mode in ("dense", "sharded") and not self.spd and table.shape[2] <= 6
        if deterministic is None:
            deterministic = det_ok and self.batch_size >= 32768
        if deterministic and not det_ok:
            raise ValueError("deterministic accumulation: dense / sharded exchange, dims <= 6")
        self.deterministic = bool(deterministic)
        self.loss = torch.zeros(1, dtype=torch.float64, device=device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.arange = torch.arange(self.batch_size, dtype=torch.int64, device=device)
        if self.deterministic:
            n = table.shape[2]
            self.rows = torch.empty(2 * self.batch_size, 2, n, n, dtype=torch.float64, device=device)
            self.partials = torch.zeros((self.batch_size + 63) // 64, 2 + n, dtype=torch.float64, device=device)
        self.capacity = 0
        self._alloc(self.batch_size)
        self.steps_loaded = 0
        self.replays = 0
        self.graphs = None
        self.key = None
        backend = dist.get_backend(group) if (dist.is_available() and dist.is_initialized()) else None
        self.capture_collective = (backend == "nccl") if capture_collective is None else bool(capture_collective)
        self.graphs_per_step = None
        # the optimiser side: one fused launch where the table qualifies (after GradientExchange: .grad are its views)
        self._extra = [p for p in self.params if p is not table]
        self._fused = None
        if self.spd:
            # the spd backward has no batch window of its own: the batch of the step is gathered from the loaded shard by two
            # index_selects on the device step counter (capturable torch ops), and the three-kernel backward gets a persistent
            # workspace (a replayed graph must not allocate)
            self.ids_b = torch.zeros(self.batch_size, 2, dtype=torch.int64, device=device)
            self.gd_b = torch.ones(self.batch_size, dtype=torch.float64, device=device)
            need = int(ops._lib.load().sympa_spd_backward_workspace_bytes(self.batch_size, table.shape[1]))
            self.spd_ws = torch.empty(need, dtype=torch.uint8, device=device) if need > 0 else None
        if not self.spd and self.mode != "sharded" and ops.FusedStep.supported(table.data) and len(self._extra) <= 2 and \
                all(p.numel() <= 64 for p in self._extra):
            self._fused = ops.FusedStep(table.data, table.grad, man.model_name, [(p.data, p.grad) for p in self._extra],
                                        counter=self.counter, projected=man.projected_counter(table.device), zero_grads=True)

    def _alloc(self, pairs):
        self.capacity = int(pairs)
        self.ids = torch.zeros(self.capacity, 2, dtype=torch.int64, device=self.device)
        self.gd = torch.ones(self.capacity, dtype=torch.float64, device=self.device)
        if self.deterministic:
            steps = max(1, self.capacity // self.batch_size)
            self.order = torch.zeros(steps, 2 * self.batch_size, dtype=torch.int32, device=self.device)
            self.rowptr = torch.zeros(steps, self.model.embeddings.embeds.shape[0] + 1, dtype=torch.int32, device=self.device)
        self.graphs = None

    def _merge_flags(self):
        return ops.FLAG_MERGE_SRC if merges_source_rows(self.model, self.deterministic, self.batch_size) else 0

    def _bwd_ws(self, table, b):
        """The persistent scratch of the split Siegel backward for batches of b pairs (None where no kernel uses one): a replayed
        graph must not allocate, so it is made once per batch size and kept."""
        cache = self.__dict__.setdefault("_bwd_ws_cache", {})
        key = (int(b), table.shape[2], table.device)
        if key not in cache:
            cache[key] = ops.siegel_backward_workspace(b, table.shape[2], self.model.manifold.model_name, table.device) \
                if table.dim() == 4 else None
        return cache[key]

    def _group_of(self, p):
        for g in self.opt.param_groups:
            if any(p is q for q in g["params"]):
                return g
        raise KeyError("parameter not in the optimiser")

    # ---- the three pieces of a step --------------------------------------------------------------------------------
    def _backward(self):
        m, man, ex = self.model, self.model.manifold, self.ex
        table = m.embeddings.embeds
        gs = m.scale.grad if m.scale.requires_grad else None
        b = self.batch_size
        if not self.accumulate_loss:
            self.loss.zero_()
        if self.spd:
            sel = self.arange + self.counter * b
            torch.index_select(self.ids, 0, sel, out=self.ids_b)
            torch.index_select(self.gd, 0, sel, out=self.gd_b)
            ops.spd_loss_backward(table.data, self.ids_b, table.grad, graph_dist=self.gd_b, scale=m.scale.data,
                                  scale_coef=m.scale_coef, loss_scale=1.0, loss=self.loss, grad_scale=gs, workspace=self.spd_ws)
            return
        wsum = man.metric.kind is MetricType.WEIGHTED_SUM
        weights = man.metric.weights if wsum else None
        gw = weights.grad if wsum else None
        if self.mode == "rows":
            ops.model_train_backward(table.data, self.ids, self.gd, b, self.loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_rows=ex.rows, step_counter=self.counter, workspace=self._bwd_ws(table, b))
            sel = self.arange + self.counter * b                 # the batch's rows of the loaded shard, on the device
            ex.idx[:b].copy_(self.ids[:, 0].index_select(0, sel))
            ex.idx[b:].copy_(self.ids[:, 1].index_select(0, sel))
        elif self.deterministic:
            ops.model_train_backward(table.data, self.ids, self.gd, b, self.loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_rows=self.rows, step_counter=self.counter, wave_partials=self.partials, workspace=self._bwd_ws(table, b),
                                     flags=self._merge_flags())
            ops.segment_sum_rows_(table.grad, self.rows, self.order, self.rowptr, step_counter=self.counter,
                                  wave_partials=self.partials, num_waves=(b + 63) // 64, partial_stride=2 + table.shape[2],
                                  loss=self.loss, grad_scale=gs, grad_weights=gw)
        else:
            ops.model_train_backward(table.data, self.ids, self.gd, b, self.loss, man.model_name, man.metric.kind.value,
                                     None if weights is None else weights.data, gw, m.scale.data, gs, m.scale_coef, 1.0,
                                     grad_table=table.grad, step_counter=self.counter, workspace=self._bwd_ws(table, b),
                                     flags=self._merge_flags())

    def _exchange(self):
        ex = self.ex
        if self.mode == "dense":
            ex.allreduce()
        elif self.mode == "rows":
            ex.exchange_rows(ex.idx[:self.batch_size], ex.idx[self.batch_size:])

    def _optimise(self):
        if self.mode == "sharded":
            self.ex.sharded_step(self.opt, self.max_grad_norm)          # leaves the flat buffer zero
            self.counter.add_(1)
            return
        if self._fused is not None:
            tg = self._group_of(self.model.embeddings.embeds)
            xg = [self._group_of(p) for p in self._extra]
            self._fused.run(tg["lr"], tg.get("weight_decay", 0.0), self.max_grad_norm,
                            [g["lr"] for g in xg], [g.get("weight_decay", 0.0) for g in xg])
            return
        self.ex.step_after_exchange(self.opt, self.max_grad_norm)
        self.ex.zero_()
        self.counter.add_(1)

    # ---- capture -----------------------------------------------------------------------------------------------------
    def _key(self):
        return (tuple((float(g["lr"]), float(g.get("weight_decay", 0.0))) for g in self.opt.param_groups), self.max_grad_norm)

    def _graph_of(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        return g

    def _capture(self):
        if ops._debug:
            raise RuntimeError("ops.set_debug(True) synchronises after every kernel and cannot run inside a hipGraph capture")
        # warm-up with lr = 0 on a side stream (allocations, status words, RCCL's lazy communicator setup): retr(x, 0) =
        # projx(x) is the identity for points on the manifold; loss, counter and gradients are put back
        saved_lr = [g["lr"] for g in self.opt.param_groups]
        saved = (self.loss.clone(), self.counter.clone())
        for g in self.opt.param_groups:
            g["lr"] = 0.0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.counter.zero_()
                self._backward()
                self._exchange()
                self._optimise()
        torch.cuda.current_stream().wait_stream(side)
        for g, lr in zip(self.opt.param_groups, saved_lr):
            g["lr"] = lr
        self.loss.copy_(saved[0])
        self.counter.copy_(saved[1])
        self.ex.zero_()
        self.graphs = None
        if (self.capture_collective or self.world == 1) and not getattr(self, "force_split", False):     # (world 1: no collective at all)
            def whole():
                self._backward()
                self._exchange()
                self._optimise()
            try:
                self.graphs = [("graph", self._graph_of(whole))]
            except Exception as e:  # noqa: BLE001 -- a backend that cannot enqueue its collective into a capture
                torch.cuda.synchronize(self.device)
                self.capture_error = repr(e)
                self.graphs = None
                self.loss.copy_(saved[0])
                self.counter.copy_(saved[1])
                self.ex.zero_()
        if self.graphs is None:
            plan = [("graph", self._graph_of(self._backward))]
            if self.mode != "sharded":
                plan.append(("eager", self._exchange))          # host-side collectives (gloo), or the scatter of the rows mode
            if self.mode == "sharded":
                plan.append(("eager", self._optimise))          # collectives inside
            else:
                plan.append(("graph", self._graph_of(self._optimise)))
            self.graphs = plan
        self.graphs_per_step = sum(1 for kind, _ in self.graphs if kind == "graph")
        self.key = self._key()

    def load_epoch(self, triplets):
        """This rank's shard of the epoch (DistributedSampler order, train.py:105-110), [T, 3] int64 on the device; returns
        the number of full batches `run_steps` may replay."""
        total = triplets.shape[0]
        if total > self.capacity:
            self._alloc(total)
        if batches_want_source_order(self.model, self.batch_size, self.deterministic):      # the order inside a batch is free
            triplets = data.sort_batches_by_source(triplets, self.batch_size)
        self.ids[:total].copy_(triplets[:, :2])
        self.gd[:total].copy_(triplets[:, 2])
        steps = total // self.batch_size
        if self.deterministic and steps > 0:       # ONE stable sort per epoch: the order the segmented sums add in
            used = self.ids[:steps * self.batch_size].view(steps, self.batch_size, 2)
            order, rowptr = ops.sorted_slots(torch.cat((used[:, :, 0], used[:, :, 1]), dim=1),
                                             self.model.embeddings.embeds.shape[0],
                                             merged_src=self.batch_size if self._merge_flags() else 0)
            self.order[:steps].copy_(order)
            self.rowptr[:steps].copy_(rowptr)
        self.counter.zero_()
        # the replayed backward ACCUMULATES into the gradient buffer and relies on the optimiser side leaving it zero: whatever
        # ran between two epochs (an eager ragged batch through the same GradientExchange) must not leak into the first step
        self.ex.zero_()
        self.steps_loaded = steps
        return self.steps_loaded

    def run_steps(self, k=None):
        if self.graphs is None or self._key() != self.key:
            self._capture()
        self.ex.check_views()
        k = self.steps_loaded if k is None else int(k)
        if k > self.steps_loaded:
            raise ValueError("more steps than full batches left in the loaded epoch")
        for _ in range(k):
            for kind, item in self.graphs:
                if kind == "graph":
                    item.replay()
                    self.replays += 1
                else:
                    item()
        if k > 0:
            ops.table_changed(self.model.embeddings.embeds)
        self.steps_loaded -= k
        return self.loss

    def reset_loss(self):
        self.loss.zero_()


def check(device):
    ops.check_status(device)
