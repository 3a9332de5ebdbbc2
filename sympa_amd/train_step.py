"""The body of the reference's training loop (sympa/runner.py:98-118: zero_grad, forward, AverageDistortionLoss,
backward, gradient clip, optimiser step) as ONE hipGraph replay per batch.

Eagerly the step is a handful of launches (memset, the fused loss+backward kernel, the norm / clip kernels of
torch.nn.utils.clip_grad_norm_, the fused RSGD kernel) issued from Python; at the reference's batch sizes (512 .. 8192
triplets) the Python and launch overhead (~0.3 ms) is several times the GPU time.  Everything in the step is
stream-ordered and allocation-free after the first call, so it is captured once per (batch size, learning rate) and
replayed: the batch is copied into static buffers, the graph runs, the loss stays on the device."""
import torch

from sympa_amd import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, batch_size, max_grad_norm, device):
        # the capture bakes every host-side scalar of the optimiser's step in as an immediate: only optimisers whose step has
        # no host state that changes from step to step may be captured (RiemannianAdam's bias corrections and step count do)
        if not getattr(optimizer, "graph_capturable", False):
            raise TypeError(f"{type(optimizer).__name__} cannot be captured in a hipGraph (its step() changes host-side state "
                            "every call); use sympa_amd.optim.RiemannianSGD or run the step eagerly")
        self.model, self.opt = model, optimizer
        self.batch_size, self.max_grad_norm = int(batch_size), float(max_grad_norm)
        self.ids = torch.zeros(self.batch_size, 2, dtype=torch.int64, device=device)
        self.gd = torch.ones(self.batch_size, dtype=torch.float64, device=device)
        self.loss = torch.zeros(1, dtype=torch.float64, device=device)
        self.graph = None
        self.key = None
        self.grad_ptrs = []
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._zero_list = None

    def _body(self):
        if self._zero_list is None:
            self.opt.zero_grad(set_to_none=False)
            self.model.fused_loss_backward(self.ids, self.gd, loss_out=self.loss)      # accumulated straight into self.loss
            self._clip_and_step()
            return
        # gradients, the loss word and the optimiser's squared-norm word zeroed by ONE multi-tensor launch
        torch._foreach_zero_(self._zero_list)
        self.model.fused_loss_backward(self.ids, self.gd, loss_out=self.loss, zero_loss_out=False)
        self.opt._sqnorm_zeroed_by_caller = True
        try:
            self._clip_and_step()
        finally:
            self.opt._sqnorm_zeroed_by_caller = False

    def _clip_and_step(self):
        if hasattr(self.opt, "clip_max_norm"):       # sympa_amd.optim.RiemannianSGD: the clip rides inside the step
            self.opt.clip_max_norm = self.max_grad_norm
            self.opt.step()
            self.opt.clip_max_norm = None
        else:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
            self.opt.step()

    def eager(self, ids, gd):
        """The same step without a graph (ragged last batch of an epoch, multi-GPU steps with an all-reduce inside)."""
        self.opt.zero_grad(set_to_none=False)
        loss = self.model.fused_loss_backward(ids, gd)
        self._clip_and_step()
        return loss

    def _key(self):
        """Everything the captured launches bake in as immediates: per-group lr and weight decay, the clip norm."""
        return (tuple((float(g["lr"]), float(g.get("weight_decay", 0.0))) for g in self.opt.param_groups),
                self.max_grad_norm)

    def _capture(self):
        if ops._debug:
            raise RuntimeError("ops.set_debug(True) synchronises after every kernel and cannot run inside a hipGraph "
                               "capture: switch it off for graphed training steps")
        # warm-up outside the capture with lr = 0: allocates the gradients / status words / foreach workspaces and
        # leaves the parameters where they are (retr(x, 0) = projx(x), the identity for points on the manifold);
        # every group gets its own lr back afterwards
        saved = [g["lr"] for g in self.opt.param_groups]
        for g in self.opt.param_groups:
            g["lr"] = 0.0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        for g, lr in zip(self.opt.param_groups, saved):
            g["lr"] = lr
        # after the warm-up every buffer exists: from here on one foreach launch zeroes them all
        self._zero_list = None
        if hasattr(self.opt, "_sqnorm_zeroed_by_caller") and hasattr(self.opt, "clip_max_norm"):
            grads = [p.grad for p in self.params if p.grad is not None]
            words = [self.loss] + [t for t in getattr(self.opt, "_sqnorm", {}).values() if t.device == self.loss.device]
            if grads and all(g.is_cuda for g in grads):
                self._zero_list = grads + words
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._body()
        self.key = self._key()
        # the graph writes the gradients through these addresses
        self.grad_ptrs = [None if p.grad is None else p.grad.data_ptr() for p in self.params]

    def __call__(self, ids, gd):
        """ids [b, 2] int64, gd [b] fp64 on the device.  Returns the batch loss as a 1-element device tensor that is
        overwritten by the next call (add it to an accumulator, do not keep it)."""
        if ids.shape[0] != self.batch_size:
            return self.eager(ids, gd)
        self.ids.copy_(ids)
        self.gd.copy_(gd)
        if self.graph is None or self._key() != self.key:
            self._capture()          # learning rate / weight decay / clip norm of any group changed (end of burn-in)
        for p, ptr in zip(self.params, self.grad_ptrs):
            if (None if p.grad is None else p.grad.data_ptr()) != ptr:
                # e.g. zero_grad(set_to_none=True) between replays: the graph would write freed memory
                raise RuntimeError("a parameter's .grad was replaced since the step was captured; keep the gradient "
                                   "tensors (zero_grad(set_to_none=False)) or build a new GraphedTrainStep")
        self.graph.replay()
        return self.loss


def check(device):
    ops.check_status(device)
