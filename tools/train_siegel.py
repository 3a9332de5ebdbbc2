#!/usr/bin/env python3
"""Minimal end-to-end harness around the MI355X path (the build's own counterpart of train.py + Runner, not a
rewrite of them: SURVEY 8f): graph -> triplets -> Model -> fused training step -> RiemannianSGD -> distortion.

    python tools/train_siegel.py --graph grid3d-125 --manifold upper --metric riem --dims 2 --epochs 50

Per batch it runs exactly two kernels: sympa_model_loss_backward (forward + AverageDistortionLoss + backward +
scatter, runner.py:101-105) and sympa_rsgd_step (geoopt RiemannianSGD, train.py:66-68), plus the gradient clip
of runner.py:115.  Multi-GPU: launch with torch.distributed.run; triplets are sharded with DistributedSampler
semantics and the gradients are exchanged over RCCL (sympa_amd/distributed.py: one flat all-reduce, touched rows, or
reduce-scatter + sharded step + all-gather); the step with the exchange in it is a graph replay too
(sympa_amd/train_step.py::DistributedTrainStep)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sympa_amd import data, ops  # noqa: E402
from sympa_amd.distributed import GradientExchange, shard_triplets  # noqa: E402
from sympa_amd.data import sort_batches_by_source  # noqa: E402
from sympa_amd.train_step import batches_want_source_order  # noqa: E402
from sympa_amd.model import Model  # noqa: E402
from sympa_amd.optim import RiemannianAdam, RiemannianSGD  # noqa: E402
from sympa_amd.train_step import DistributedTrainStep, GraphedTrainStep  # noqa: E402


def evaluate(model, ids, gd, batch):
    """Average distortion |d_manifold - d_graph| / d_graph (sympa/metrics.py:21, runner.py:124-135): Model.evaluate hands
    the batches of the split to the fused multi-batch kernel as one list (one C call per evaluation)."""
    return model.evaluate(ids, gd, batch)


def train(args, log=print):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(args.seed)
    trip, id2node = data.graph_triplets(data.named_graph(args.graph))
    args.num_points = len(id2node)
    model = Model(args).to(dev)
    if args.optim == "radam":        # train.py:69-70
        opt = RiemannianAdam(model.parameters(), lr=args.learning_rate * world, eps=1e-7, stabilize=None)
    else:                            # train.py:66-68
        opt = RiemannianSGD(model.parameters(), lr=args.learning_rate * world, weight_decay=0.0, stabilize=None)
    ids_all = trip[:, :2].contiguous().to(dev)
    gd_all = trip[:, 2].to(torch.float64).to(dev)
    batch = max(1, args.batch_size // world)
    history = []
    # single GPU: the whole step is one hipGraph replay
    graphed = GraphedTrainStep(model, opt, batch, args.max_grad_norm, dev,
                               deterministic=True if args.deterministic else None, accumulate_loss=True) \
        if (world == 1 and args.graph_step and args.grad_exchange == "none") else None
    # (RiemannianAdam keeps its powers b^t in device words, so its step is captured too -- as the same two kernels where the
    # table qualifies: sympa_radam_step_fused)
    # N > 1 (or --grad_exchange given): gradients live in one persistent flat buffer; the table gradient travels dense
    # (one in-place all-reduce) or as touched rows (all-gather of the 2 b per-pair rows), whichever message is smaller
    ex = None
    dstep = None
    if world > 1 or args.grad_exchange != "none":
        mode = "auto" if args.grad_exchange == "none" else args.grad_exchange
        if args.graph_step and args.optim == "rsgd" and args.manifold in ("upper", "bounded", "spd") and mode != "rows":
            # round 4: the step with the exchange in the middle is replayed too -- backward graph, the collective on the same
            # stream (captured inside the graph where the backend enqueues it: RCCL), optimiser graph; an epoch's shard is
            # loaded once and addressed through the device step counter
            dstep = DistributedTrainStep(model, opt, batch, args.max_grad_norm, dev, mode=mode, accumulate_loss=True)
            ex = dstep.ex
        else:
            ex = GradientExchange(list(model.parameters()), table=model.embeddings.embeds, local_batch=batch, mode=mode)
        if rank == 0:
            log(f"gradient exchange: {ex.mode}, {ex.message_bytes / 1e6:.3f} MB sent per rank per step"
                + ("" if dstep is None else ", replayed graphs"))
    # steps with an exchange in the middle run eagerly, but the optimiser side is still ONE launch where the fused kernel
    # applies (clip norm + RiemannianSGD + scale step + zero_grad on the exchanged gradient, which lives in ex's flat buffer)
    stepper = None
    if ex is not None and dstep is None:
        stepper = GraphedTrainStep(model, opt, batch, args.max_grad_norm, dev)
        if stepper.mode == "two_kernels":
            stepper._ensure_fused()          # after GradientExchange: p.grad are views of its flat buffer
        else:
            stepper = None
    for epoch in range(1, args.epochs + 1):
        mine = shard_triplets(trip, rank, world, epoch=epoch, seed=0).to(dev)
        # inside every batch: pairs with the same source row adjacent (free for SGD) -- load_epoch of the replayed steps sorts
        # by itself; only the eager per-batch loop needs it here (one argsort per epoch, not two)
        loads_epoch = dstep is not None or (graphed is not None and graphed.mode == "two_kernels")
        if not loads_epoch and batches_want_source_order(model, batch):
            mine = sort_batches_by_source(mine, batch)
        t0 = time.perf_counter()
        lr = args.learning_rate * world / (10.0 if epoch < args.burnin else 1.0)   # runner.py:162-170
        for g in opt.param_groups:
            g["lr"] = lr
        loss_sum = torch.zeros(1, dtype=torch.float64, device=dev)
        first = 0
        if graphed is not None:
            graphed.reset_loss()
            if graphed.mode == "two_kernels":
                # the epoch's triplets are loaded once; every full batch is one replay of a two-kernel graph that finds
                # its batch through a device counter (no per-step copy, memset or host arithmetic)
                first = graphed.load_epoch(mine) * batch
                graphed.run_steps()
        if dstep is not None:
            dstep.reset_loss()
            first = dstep.load_epoch(mine) * batch
            dstep.run_steps()                 # recaptures when the learning rate changed (end of burn-in)
        for s in range(first, mine.shape[0], batch):
            b = mine[s:s + batch]
            if graphed is not None:
                graphed(b[:, :2], b[:, 2].to(torch.float64))        # accumulates into graphed.loss
                continue
            ids, gd = b[:, :2].contiguous(), b[:, 2].to(torch.float64)
            if ex is None:
                opt.zero_grad(set_to_none=False)
                loss_sum += model.fused_loss_backward(ids, gd)
            elif ex.mode == "sharded":
                ex.zero_()
                loss_sum += model.fused_loss_backward(ids, gd)
                ex.sharded_step(opt, args.max_grad_norm)
                continue
            else:
                ex.zero_()
                if ex.mode == "rows" and ids.shape[0] == batch:
                    loss_sum += model.fused_loss_backward_rows(ids, gd, ex.rows)
                    ex.exchange_rows(ids[:, 0], ids[:, 1])
                else:                                   # dense mode, or the ragged last batch of an epoch
                    loss_sum += model.fused_loss_backward(ids, gd)
                    ex.allreduce()
            if stepper is not None:
                stepper._fused_step()                                                       # runner.py:115-118 in one launch
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), args.max_grad_norm)      # runner.py:115
                opt.step()
        # the reference asserts inside every dist() call (siegel_manifold.py:64-66) and checks all points once per epoch
        # (runner.py:180-184); here the status word of the kernels is read once per epoch (one host sync)
        ops.check_status(dev)
        if graphed is not None:
            loss_sum += graphed.loss
        if dstep is not None:
            loss_sum += dstep.loss
        if epoch % args.val_every == 0 or epoch == args.epochs:
            torch.cuda.synchronize(dev)
            t_train = time.perf_counter() - t0
            distortion = evaluate(model, ids_all, gd_all, args.batch_size)
            ops.check_status(dev)
            history.append((epoch, float(loss_sum) / max(1, mine.shape[0]), distortion))
            if rank == 0:
                log(f"epoch {epoch:4d}  loss/triplet {history[-1][1]:.4f}  avg distortion {distortion:.4f}  "
                    f"{t_train * 1e3:.1f} ms/epoch ({mine.shape[0] / t_train / 1e6:.2f} M triplets/s)  "
                    f"projected {model.manifold.projected_points}")
    return model, history


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="grid3d-125")
    ap.add_argument("--manifold", default="upper")
    ap.add_argument("--metric", default="riem")
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--scale_init", type=float, default=1.0)
    ap.add_argument("--scale_coef", type=float, default=1.0)
    ap.add_argument("--train_scale", action="store_true", default=False)
    ap.add_argument("--learning_rate", type=float, default=1e-2)
    ap.add_argument("--optim", default="rsgd", choices=["rsgd", "radam"])
    ap.add_argument("--max_grad_norm", type=float, default=50.0)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--burnin", type=int, default=10)
    ap.add_argument("--val_every", type=int, default=5)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--grad_exchange", default="none", choices=["none", "auto", "dense", "rows", "sharded"],
                    help="single GPU: run the step through sympa_amd.distributed.GradientExchange anyway (tests); with "
                         "N > 1 GPUs the exchange is always on and this picks its mode (none = auto)")
    ap.add_argument("--deterministic", action="store_true", default=False,
                    help="graphed two-kernel step only: per-pair gradient rows + a segmented sum in a precomputed order "
                         "instead of the fp64-atomic scatter, fixed-order sums for the loss and the scale gradient: two "
                         "runs give the same bits.  Without the flag the deterministic form is still taken where it is the "
                         "faster one (batches of 32 768 triplets and more)")
    ap.add_argument("--no_graph_step", dest="graph_step", action="store_false", default=True,
                    help="launch the kernels of a step one by one instead of replaying one hipGraph per batch")
    return ap


if __name__ == "__main__":
    train(parser().parse_args())
