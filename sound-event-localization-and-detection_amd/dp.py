"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device; this is new design, not a translation.  Per step there is
exactly ONE collective: an all-reduce (sum) of FlatAdam's flat gradient buffer (6.5 MB for DQ-8ch), the
1/world scaling is folded into the Adam kernel's `grad_scale`.  At these sizes a ring over the xGMI mesh
is latency-bound, so the payload is never split into per-tensor or per-bucket calls.  BatchNorm uses
local (per-rank) batch statistics, as nn.DataParallel -- the only multi-GPU mode the reference hints at
(train.py:27,49) -- would; running statistics are averaged across ranks on request.

Works on CPU tensors with the gloo backend too (that is how the tests cover world_size 2 here).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_batch(global_batch, rank=None, world=None):
    """Even split of the global minibatch on dim 0: returns (start, stop) of this rank's slice."""
    world = world_size() if world is None else world
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class FlatGradSync:
    """Gradient exchange for any optimiser that exposes `flat_grad` (FlatAdam) -- or, for plain tensors,
    a flat buffer built once from a parameter list (used by the CPU/gloo tests)."""

    def __init__(self, flat_grad=None, params=None, group=None):
        self.group = group
        self.params = None
        if flat_grad is not None:
            self.flat = flat_grad
        else:
            self.params = [p for p in params if p.requires_grad]
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)

    def pack(self):
        """Only for the plain-parameter mode: copy .grad tensors (missing ones count as zero) into the buffer."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def unpack(self, scale=1.0):
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view(p.shape) * scale
            p.grad = g.clone()
            off += n

    def all_reduce(self, async_op=False):
        """Sum over ranks, in place.  Returns the work handle when async_op (overlap with the tail of backward)."""
        if world_size() == 1:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def average_scale(self):
        return 1.0 / world_size()


def late_parameters(model):
    """Parameters whose gradients a backward pass completes LAST: the front-end convolution stages
    (`ConvTC_Block.cnn`, model.py:261-287) of every branch -- <1 % of the gradient bytes.  Pass them to
    `FlatAdam(..., late=...)` so that they form the small bucket at the front of the flat gradient buffer."""
    out = []
    for m in model.modules():
        if type(m).__name__ == "ConvTC_Block":
            out += list(m.cnn.parameters())
    return out


def cut_backward_here(module, x):
    """Called by a module's forward at the point where a `BackwardCut` splits the backward pass: returns `x` itself when
    no cut is installed on `module`, else a detached leaf that stands in for it (the pair is remembered for `finish`)."""
    cuts = getattr(module, "_backward_cuts", None)
    if cuts is None or not x.requires_grad:
        return x
    if any(leaf.grad is not None for _, leaf in cuts):
        # a backward pass ran to the cut and nobody ran the rest: the front end would silently get no gradient
        raise RuntimeError("BackwardCut: the previous step's backward pass stopped at the front-end cut and finish() was "
                           "never called (use dp.dp_train_step / GraphedTrainStep, or remove() the cut before a plain "
                           "loss.backward())")
    leaf = x.detach().requires_grad_(True)
    cuts.append((x, leaf))
    return leaf


class BackwardCut:
    """Splits the backward pass at the input of every TCN (the output of the front-end convolutions): `loss.backward()`
    then stops there -- every gradient of the TCN, the attention and the heads is complete and can be exchanged -- and
    `finish()` runs the rest (the front end), which overlaps that exchange.  The model cooperates through
    `cut_backward_here` (model.ConvTC_Block.forward calls it on its TCN input)."""

    def __init__(self, model):
        self.cuts = []
        self.blocks = [m for m in model.modules() if type(m).__name__ == "ConvTC_Block"]
        for b in self.blocks:
            b._backward_cuts = self.cuts

    def reset(self):
        self.cuts.clear()

    def finish(self):
        pending = [(a, leaf.grad) for a, leaf in self.cuts if leaf.grad is not None]
        self.cuts.clear()
        if pending:
            torch.autograd.backward([a for a, _ in pending], [g for _, g in pending])

    def remove(self):
        for b in self.blocks:
            if getattr(b, "_backward_cuts", None) is self.cuts:
                del b._backward_cuts

    def install(self):
        for b in self.blocks:
            b._backward_cuts = self.cuts

    def __enter__(self):
        self.install()
        self.reset()
        return self

    def __exit__(self, *exc):
        self.remove()
        return False


class BucketedGradSync:
    """The data-parallel gradient exchange of a FlatAdam built with `late=late_parameters(model)`: two all-reduces per
    step over contiguous ranges of the one flat gradient buffer,

        main bucket  flat_grad[late_numel:]   (TCN, attention, heads: > 99 % of the bytes) -- issued asynchronously as
                                              soon as the backward pass has reached the front-end cut, so it runs on
                                              RCCL's stream while the front-end convolutions' backward runs on ours;
        late bucket  flat_grad[:late_numel]   (front-end convolutions) -- issued when the backward pass has ended.

    Each payload is a few MB at most, latency-bound on the xGMI mesh, so neither is split further.  `wait()` makes the
    compute stream wait for both; the 1/world scale is folded into the Adam kernel (`average_scale`)."""

    def __init__(self, optimizer, model=None, group=None):
        self.group = group
        self.flat = optimizer.flat_grad
        self.late_numel = int(getattr(optimizer, "late_numel", 0))
        self.world = world_size()
        self.cut = BackwardCut(model) if (model is not None and self.world > 1 and self.late_numel > 0) else None
        self._work = []

    def _reduce(self, t):
        if self.world > 1 and t.numel() > 0:
            self._work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_main(self):
        self._reduce(self.flat[self.late_numel:])

    def reduce_late(self):
        self._reduce(self.flat[:self.late_numel])

    def all_reduce(self, async_op=False):
        """Whole buffer at once (no overlap): the round-1 interface."""
        self.reduce_main()
        self.reduce_late()
        if not async_op:
            self.wait()

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def average_scale(self):
        return 1.0 / self.world


def broadcast_parameters(module_or_flat, src=0):
    """Make every rank start from rank `src`'s weights (and buffers)."""
    if world_size() == 1:
        return
    if torch.is_tensor(module_or_flat):
        dist.broadcast(module_or_flat, src)
        if module_or_flat.is_cuda:          # parameters are views of this buffer: their packed forms are stale
            from . import hip_ops as H
            H.hcq_weights.weights_changed()
        return
    for t in list(module_or_flat.parameters()) + list(module_or_flat.buffers()):
        dist.broadcast(t.data, src)
    from . import hip_ops as H              # written through .data: no version counter moved
    H.hcq_weights.weights_changed()


def average_bn_running_stats(module):
    """Optional: average BatchNorm running_mean / running_var across ranks (e.g. before checkpointing)."""
    w = world_size()
    if w == 1:
        return
    bufs = [b for n, b in module.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat)
    flat /= w
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


def _join_side_stream():
    """The accumulating weight-gradient kernels run on hip_ops' side stream: a collective may only read the gradient
    buffer after the compute stream has joined it."""
    from . import hip_ops as H
    H.join_side_stream()


def dp_train_step(model, optimizer, sync, x, target, n_sed, loss_fn, sed_weight=1.0, doa_weight=5.0):
    """One data-parallel step on this rank's shard (eager; train.GraphedTrainStep is the recorded form):
    zero_grad -> fwd -> loss -> bwd [-> exchange of the main bucket || bwd of the front end -> exchange of the late
    bucket] -> Adam(mean gradient)."""
    optimizer.zero_grad()
    cut = getattr(sync, "cut", None)
    if cut is not None:
        cut.reset()
    sed, doa = model(x)
    loss = loss_fn(sed, doa, target, n_sed, sed_weight, doa_weight)
    from .hip_ops import backward_from_loss
    backward_from_loss(loss)
    if x.is_cuda:
        _join_side_stream()
    if isinstance(sync, BucketedGradSync):
        sync.reduce_main()
        if cut is not None:
            cut.finish()
            if x.is_cuda:
                _join_side_stream()
        sync.reduce_late()
        sync.wait()
    else:
        sync.all_reduce()
    optimizer.step(grad_scale=sync.average_scale())
    return loss


def seed_rank_streams(rank):
    """Ranks must draw different dropout masks for their different shards (weights stay identical: they are broadcast).
    Folds the rank into the key of hip_ops' counter-based RNG."""
    from . import hip_ops as H
    H.philox.stream_id = int(rank)
