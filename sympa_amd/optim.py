"""RiemannianSGD for the Siegel models.

The reference builds `geoopt.optim.RiemannianSGD(params, lr, weight_decay, stabilize=None)` (train.py:66-68);
geoopt is an un-vendored dependency absent from the reference tree, so its step is restated from the
published algorithm (geoopt/optim/rsgd.py, momentum = 0):
    grad <- grad + weight_decay * point;  grad <- manifold.egrad2rgrad(point, grad)
    point <- manifold.retr(point, -lr * grad)
For a table on a Siegel manifold the whole step is ONE HIP kernel over the rows (sympa_rsgd_step);
parameters without a manifold (the model scale, the wsum weights) get the Euclidean update."""
import torch

from sympa_amd import ops
from sympa_amd.manifolds.siegel_manifold import SiegelManifold
from sympa_amd.manifolds.spd import SymmetricPositiveDefinite


class RiemannianSGD(torch.optim.Optimizer):
    graph_capturable = True      # step() keeps no host-side state that changes between calls (sympa_amd.train_step)

    def __init__(self, params, lr, weight_decay=0.0, stabilize=None):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))
        self._stabilize = stabilize   # accepted for signature compatibility; Siegel retr already projects
        # Extension: fold torch.nn.utils.clip_grad_norm_(params, max_norm) (runner.py:115) into the step.  The total
        # norm is accumulated on the device and the factor min(1, max_norm / (norm + 1e-6)) is applied to the
        # gradients as the update kernels read them; .grad itself is left unscaled.
        self.clip_max_norm = None
        self._sqnorm = {}
        self._sqnorm_zeroed_by_caller = False      # GraphedTrainStep zeroes it together with the gradients (one launch)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        sq = coef = None
        if self.clip_max_norm is not None:
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grads and all(t.is_cuda and t.dtype == torch.float64 for t in grads):
                key = str(grads[0].device)
                if key not in self._sqnorm:
                    self._sqnorm[key] = torch.zeros(1, dtype=torch.float64, device=grads[0].device)
                sq = self._sqnorm[key]
                if not self._sqnorm_zeroed_by_caller:
                    sq.zero_()
                for t in grads:
                    ops.sqnorm_accum_(t if t.is_contiguous() else t.contiguous(), sq)
            else:
                torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"]], self.clip_max_norm)
        for group in self.param_groups:
            lr, wd = group["lr"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                manifold = getattr(p, "manifold", None)
                if manifold is not None:
                    ops.table_changed(p)       # the kernels below write through p.data: torch's version counter does not see them
                if isinstance(manifold, SymmetricPositiveDefinite) and p.is_cuda:
                    if p.dtype != torch.float64:
                        raise TypeError("the spd table must be float64 (config.py:17-18)")
                    # geoopt's SPD step: retr(x, -lr x sym(g + wd x) x), one kernel over the table
                    ops.spd_rsgd_step_(p.data, p.grad, lr, wd, clip_sqnorm=sq,
                                       max_norm=self.clip_max_norm if sq is not None else None)
                elif isinstance(manifold, SiegelManifold) and p.is_cuda and sq is not None:
                    ops.rsgd_step_(p.data, p.grad, manifold.model_name, lr, wd, counter=manifold.projected_counter(p.device),
                                   clip_sqnorm=sq, max_norm=self.clip_max_norm)
                elif isinstance(manifold, SiegelManifold) and p.is_cuda:
                    # the kernel adds the number of projected rows to a persistent device counter (read lazily by
                    # manifold.projected_points): no allocation, no synchronisation, graph-capturable
                    ops.rsgd_step_(p.data, p.grad, manifold.model_name, lr, wd, counter=manifold.projected_counter(p.device))
                elif (p.is_cuda and p.dtype == torch.float64 and p.grad.dtype == torch.float64 and p.is_contiguous()
                      and p.grad.is_contiguous()):
                    # a parameter without a manifold (the scale): one kernel, clip folded in
                    ops.sgd_step_clipped_(p.data, p.grad, lr, wd, clip_sqnorm=sq,
                                          max_norm=self.clip_max_norm if sq is not None else None)
                else:
                    g = p.grad
                    if sq is not None:
                        if coef is None:
                            coef = (self.clip_max_norm / (sq.sqrt() + 1e-6)).clamp(max=1.0)
                        g = g * coef.to(g.dtype)
                    if wd != 0:
                        g = g.add(p, alpha=wd)
                    p.add_(g, alpha=-lr)
        return loss


class RiemannianAdam(torch.optim.Optimizer):
    """geoopt.optim.RiemannianAdam for the Siegel models (train.py:69-70: `--optim radam`, lr, eps=1e-7, stabilize=None;
    geoopt is absent from the reference tree, the step is restated from geoopt/optim/radam.py):
        g  <- egrad2rgrad(x, grad + weight_decay * x)
        m  <- b1 m + (1 - b1) g
        v  <- b2 v + (1 - b2) inner(x, g, g)              (component_inner: one value per point, broadcast)
        x  <- retr(x, -lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps));   m is transported by the identity
    (SiegelManifold.transp returns the vector unchanged, siegel_manifold.py:142-154).  egrad2rgrad, inner and retr
    (= projx(x + u)) are HIP kernels over the table rows; the moment updates are elementwise torch ops on the device.
    Parameters without a manifold get the ordinary Adam update.  amsgrad is not built.

    The powers b1^t, b2^t live in two device words per parameter and are advanced by the step itself, so nothing the step
    computes depends on host state that changes from call to call: the step can be captured in a hipGraph
    (sympa_amd.train_step.GraphedTrainStep, classic mode) like RiemannianSGD's.  group["step"] still counts the host-side
    calls, for information only."""
    graph_capturable = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, stabilize=None):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, step=0))
        self._stabilize = stabilize

    def _init_param_state(self, p, betas, steps_taken=0):
        """Creates whatever the state of `p` lacks (all of it on first use; `bias_pows` / `betas` when the state came from
        a checkpoint of the format that kept the powers on the host) and keeps the device copy of the betas equal to the
        group's: geoopt evaluates betas ** step from the live group, so a change of group["betas"] takes effect in the
        moments AND in the bias corrections from the next step on (the powers restart from the new betas at the same t)."""
        state = self.state[p]
        manifold = getattr(p, "manifold", None)
        siegel = isinstance(manifold, SiegelManifold) and p.is_cuda
        if "exp_avg" not in state:
            state["exp_avg"] = torch.zeros_like(p)
        if "exp_avg_sq" not in state:
            state["exp_avg_sq"] = torch.zeros(p.shape[0], dtype=p.dtype, device=p.device) if siegel else torch.zeros_like(p)
        b = (float(betas[0]), float(betas[1]))
        if "bias_pows" not in state:                       # (b1^t, b2^t), advanced by the step itself
            t = int(steps_taken)
            state["bias_pows"] = torch.tensor([b[0] ** t, b[1] ** t], dtype=torch.float64, device=p.device)
        if "betas" not in state:
            state["betas"] = torch.tensor(b, dtype=torch.float64, device=p.device)
            state["betas_host"] = b
        elif state.get("betas_host") != b:
            if "betas_host" in state or tuple(float(x) for x in state["betas"].tolist()) != b:
                # in place: captured graphs hold these addresses.  b_new^t from b_old^t: t = log(pow) / log(b_old)
                old = state["betas"].clone()
                t = torch.log(state["bias_pows"]) / torch.log(old)
                state["betas"].copy_(torch.tensor(b, dtype=torch.float64, device=p.device))
                state["bias_pows"].copy_(torch.pow(state["betas"], torch.nan_to_num(t.round(), nan=0.0)))
            state["betas_host"] = b
        return state

    def init_state(self):
        """Creates the moment buffers and the power words of every parameter (normally done by the first step)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    self._init_param_state(p, group["betas"], group["step"])

    def snapshot_state(self):
        """Copies of every state tensor and step count: GraphedTrainStep's warm-up steps (lr = 0) must not count."""
        return ([g["step"] for g in self.param_groups],
                {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.state.items()})

    def restore_state(self, snap):
        steps, states = snap
        for g, n in zip(self.param_groups, steps):
            g["step"] = n
        for p, st in states.items():
            for k, v in st.items():
                self.state[p][k].copy_(v)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            group["step"] += 1
            for p in group["params"]:
                if p.grad is None:
                    continue
                manifold = getattr(p, "manifold", None)
                siegel = isinstance(manifold, SiegelManifold) and p.is_cuda
                if manifold is not None:
                    ops.table_changed(p)       # (see RiemannianSGD.step)
                if isinstance(manifold, SymmetricPositiveDefinite):
                    raise NotImplementedError("RiemannianAdam on the spd model needs geoopt's parallel transport: use rsgd")
                state = self._init_param_state(p, (b1, b2), group["step"] - 1)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                pows = state["bias_pows"].mul_(state["betas"])      # (b1^t, b2^t), t = steps this parameter has taken: one launch
                if siegel and p.shape[2] <= ops.RADAM_FUSED_MAX_DIMS and p.data.is_contiguous() and p.dtype == torch.float64:
                    # the whole row update in one kernel (C-ABI sympa_radam_step); the powers were advanced just above
                    ops.radam_step_(p.data, p.grad, m, v, state["bias_pows"], manifold.model_name, lr, (b1, b2), eps, wd,
                                    counter=manifold.projected_counter(p.device))
                    continue
                g = p.grad
                if wd != 0:
                    g = g.add(p, alpha=wd)
                bc = pows.neg().add_(1.0).to(p.dtype)               # bias corrections 1 - b^t
                bc1, bc2 = bc[0], bc[1]
                if siegel:
                    g = ops.egrad2rgrad(p.data, g, manifold.model_name)
                    m.mul_(b1).add_(g, alpha=1.0 - b1)
                    v.mul_(b2).add_(ops.tangent_sqnorm(p.data, g, manifold.model_name), alpha=1.0 - b2)
                    denom = v.div(bc2).sqrt_().add_(eps)
                    direction = m.div(bc1) / denom.view(-1, 1, 1, 1)
                    counter = manifold.projected_counter(p.device)
                    p.data.copy_(ops.projx(p.data.add(direction, alpha=-lr), manifold.model_name, counter=counter))
                else:
                    m.mul_(b1).add_(g, alpha=1.0 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                    denom = v.div(bc2).sqrt_().add_(eps)
                    p.add_(m.div(bc1) / denom, alpha=-lr)
        return loss
