"""Training harness with the CLI of the reference's train.py (`python train.py --TextArgs=config/X.txt`,
same flag names and defaults, train.py:719-818) and its step semantics (train.py:186-204, 498-508, 538-571):

    zero_grad -> forward -> BCELoss(sed) * w_sed + MSELoss(doa) * w_doa -> backward -> Adam(lr 1e-4)

on the gfx950 kernels: the loss is one fused kernel, Adam is ONE launch over a flat parameter buffer,
and under torch.distributed (one process per GPU, RCCL) the gradient exchange is ONE all-reduce of the
flat gradient buffer (the reference is single-device).

The reference's own `main()` cannot run on a current stack (np.Inf, StepLR(verbose=), wandb, ...);
this file restates its behaviour rather than its text.
"""
import argparse
import ast
import os
import pickle
import shutil
import sys
import time

import numpy as np
import torch
import torch.nn as nn

from . import hip_ops as H
from .model import SELD_Model


# ------------------------------------------------------------------------------------------
# loss / optimiser
# ------------------------------------------------------------------------------------------
def seld_loss_fn(sed, doa, target, n_sed, sed_weight=1.0, doa_weight=5.0):
    """BCE(sed, target[..., :n_sed]) * sed_weight + MSE(doa, target[..., n_sed:]) * doa_weight, both means."""
    assert target.shape[-1] == sed.shape[-1] + doa.shape[-1] and sed.shape[-1] == n_sed
    return H.seld_loss(sed, doa, target, sed_weight, doa_weight)


class BCELoss(nn.Module):
    """Marker modules so that `seld_loss(x, target, model, criterion_sed, criterion_doa)` keeps the reference's
    signature; the two criteria are evaluated together by one fused kernel."""


class MSELoss(nn.Module):
    pass


def seld_loss(x, target, model, criterion_sed, criterion_doa, args=None):
    """train.py:186-204.  `args` carries output_classes / class_overlaps / loss weights (the reference reads a
    module-level global)."""
    a = args if args is not None else globals().get("args")
    n_sed = int(a.output_classes * a.class_overlaps)
    sed, doa = model(x)
    if isinstance(criterion_sed, BCELoss) and isinstance(criterion_doa, MSELoss):
        return seld_loss_fn(sed, doa, target, n_sed, a.sed_loss_weight, a.doa_loss_weight)
    t_sed, t_doa = target[:, :, :n_sed], target[:, :, n_sed:]
    return criterion_sed(torch.flatten(sed, 1), torch.flatten(t_sed, 1)) * a.sed_loss_weight + \
        criterion_doa(torch.flatten(doa, 1), torch.flatten(t_doa, 1)) * a.doa_loss_weight


class FlatAdam:
    """torch.optim.Adam semantics (train.py:502) with every parameter, gradient and moment living in one
    flat fp32 buffer each: `step()` is a single kernel launch and a data-parallel gradient exchange is a
    single all-reduce of `flat_grad` (see dp.py).  Parameters that never receive a gradient (SURVEY App. B:
    batch_gate1.*, the last block's conv2_residual) keep a zero gradient, which leaves them unchanged, as
    torch.optim.Adam does by skipping them."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, late=None):
        """`late`: parameters whose gradients are the LAST ones a backward pass completes (the front-end convolutions,
        see dp.late_parameters).  They are stored at the front of the flat buffers, so [0, late_numel) and
        [late_numel, total) are the two contiguous buckets of the data-parallel exchange (dp.BucketedGradSync: the big
        bucket is all-reduced while the front end's backward pass is still running).  `self.params`, the state dict
        and the relative order inside each bucket keep `model.parameters()` order."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        late_ids = {id(p) for p in (late or ())}
        storage = [p for p in self.params if id(p) in late_ids] + [p for p in self.params if id(p) not in late_ids]
        self.late_numel = sum(p.numel() for p in self.params if id(p) in late_ids)
        self.flat_param = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        self._offset = {}
        off = 0
        with torch.no_grad():
            for p in storage:
                n = p.numel()
                self._offset[id(p)] = off
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                p._seld_direct_grad = True      # HIP backward kernels accumulate straight into this view
                p._seld_owner = self            # ... and may reduce into a slot that is still zero (grad_generation)
                off += n
        self.offsets = [self._offset[id(p)] for p in self.params]      # flat offset of params[i]
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=self.params)]
        self.step_count = 0
        self.grad_generation = 0

    def zero_grad(self, set_to_none=False, state=None):
        """`state`: the device-resident step state (hip_ops.philox.state): also advance it (graph-recorded steps)."""
        H.deferred_wgrads.discard()                  # jobs of a backward pass nobody finished would add into the fresh buffer
        if self.flat_grad.is_cuda:
            H.step_begin(self.flat_grad, state)      # one launch: zero fill (+ step state)
        else:
            self.flat_grad.zero_()
        self.grad_generation += 1       # every slot is zero again (hip_ops._claim_grad_slots)
        for p, off in zip(self.params, self.offsets):           # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def step(self, grad_scale=1.0, state=None):
        """`state`: take the step number and the learning rate from the device-resident step state instead of the
        host's (graph-recorded steps; GraphedTrainStep keeps the host's count in step and `sync_from_host` pushes host-side
        changes -- eager steps, a restored optimiser -- back into the state before the next replay)."""
        H.deferred_wgrads.flush()       # grouped weight gradients still pending (a backward pass outside the autograd engine)
        H.join_side_stream()            # weight gradients issued on the side stream (hip_ops._on_side_stream)
        H.hcq_weights.weights_changed() # the packed weight forms of the fast-product convolutions are stale now
        g = self.param_groups[0]
        if state is not None:
            H.adam_flat_step_state(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, state, g["betas"][0],
                                   g["betas"][1], g["eps"], g["weight_decay"], grad_scale)
            return
        self.step_count += 1
        H.adam_flat_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, g["lr"],
                         g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], grad_scale)

    def state_dict(self):
        """torch.optim.Adam's layout (the reference saves `optimizer.state_dict()`, train.py:36): per-parameter
        `step / exp_avg / exp_avg_sq` keyed by the parameter's index in `model.parameters()` order plus one
        param group with torch's hyper-parameter keys, so the reference's `load_model` accepts our checkpoints.
        Every parameter gets an entry once a step was taken (torch omits parameters that never had a gradient;
        an extra entry is ignored by torch for as long as the parameter has no gradient)."""
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[off:off + n].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[off:off + n].view(p.shape).clone())
        group = dict(_torch_adam_defaults())
        group.update({k: v for k, v in self.param_groups[0].items() if k != "params"})
        group["params"] = list(range(len(self.params)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (reference checkpoints) and this class's round-1 flat layout."""
        H.hcq_weights.weights_changed()      # a restore usually comes with new weights: never serve forms from before it
        if "exp_avg" in sd:                                   # flat layout
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups[0].update(sd["param_groups"][0])
            return
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        index_of = {pid: i for i, pid in enumerate(groups[0]["params"])}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        offsets = self.offsets
        steps = set()
        for pid, st in sd["state"].items():
            i = index_of[pid]
            p, o = self.params[i], offsets[i]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
            self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a checkpoint of one Adam over the model")
        self.step_count = steps.pop() if steps else 0
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        if groups[0].get("amsgrad") or groups[0].get("maximize"):
            raise ValueError("amsgrad / maximize checkpoints are not supported (the reference never sets them)")


class GraphedTrainStep:
    """One training step -- zero_grad -> forward -> BCE + w*MSE -> backward -> [gradient exchange] -> Adam
    (train.py:552-560) -- recorded ONCE as HIP graphs and replayed: ~400 kernel launches on two streams become one
    host call, so the step costs what its kernels cost and nothing of the Python / autograd / ctypes time between them
    (at 16 samples per GPU the eager step is host-bound).  Nothing is traced or compiled: the recorded launches are the
    same C-ABI calls the eager step makes.

    What a replay must not freeze is kept in device memory and read by the kernels: the Philox base of the dropout
    masks, the optimiser's step number and the learning rate (the 4-word step state of seld_step_begin).

    Single process: one graph.  Data parallel (`sync` = dp.BucketedGradSync): three graphs with the two collectives
    issued eagerly between them, so that no collective is ever captured:
        A  zero_grad, forward, loss, backward down to the front-end cut    -> all-reduce of the big bucket (async)
        B  backward of the front-end convolutions (overlaps that all-reduce) -> all-reduce of the late bucket
        C  Adam on the averaged gradients
    The batch is static: `__call__(x, target)` copies into the recorded input buffers (same shapes)."""

    def __init__(self, model, optimizer, x, target, n_sed, sed_weight=1.0, doa_weight=5.0, sync=None, warmup=2):
        self.model, self.opt, self.sync = model, optimizer, sync
        self.n_sed, self.w = n_sed, (float(sed_weight), float(doa_weight))
        self.x, self.target = x.clone(), target.clone()
        self.state = H.philox.state(x.device)
        self.cut = sync.cut if (sync is not None and getattr(sync, "cut", None) is not None and sync.world > 1) else None
        self._lr = None
        dev = x.device
        # eager warm-up on a side stream (allocator pools, kernel modules, host-side caches), as torch recommends
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        # the step state takes over from the host counters
        self.state[1] = self.opt.step_count
        self.state[0] = H.philox.offset
        self.state[3] = 0
        H.philox.offset = 0
        self._push_lr()
        H.hcq_weights.refresh_table()       # host-to-device copies of the weight-form table: not allowed while capturing
        self.graphs = []
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self.loss = self._phase_a()
            if self.cut is None:
                self._phase_b()
                self._phase_c()
        self.graphs.append(ga)
        if self.cut is not None:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool()):
                self._phase_b()
            gc_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gc_, pool=ga.pool()):
                self._phase_c()
            self.graphs += [gb, gc_]
        # the recording itself ran no kernel: undo its host-side bookkeeping, publish the draws per step
        self.state[3] = H.philox.offset
        H.philox.offset = 0
        self.replays = 0
        self._step_seen = self.opt.step_count
        # the recorded weight-form launch reads the cache's table / block starts / form buffers through raw pointers:
        # hold them, the cache builds NEW tensors when a shape is registered later (hip_ops._HcqWeights.pin_for_graph)
        self._hcq_pinned = H.hcq_weights.pin_for_graph()

    def _push_lr(self):
        lr = float(self.opt.param_groups[0]["lr"])
        if lr != self._lr:
            self._lr = lr
            self.state[2] = int(np.float32(lr).view(np.uint32))

    def _eager(self):
        self.opt.zero_grad()
        if self.cut is not None:
            self.cut.reset()
        sed, doa = self.model(self.x)
        loss = seld_loss_fn(sed, doa, self.target, self.n_sed, *self.w)
        H.backward_from_loss(loss)
        if self.cut is not None:
            self.cut.finish()
        H.join_side_stream()
        if self.sync is not None:
            self.sync.reduce_main()
            self.sync.reduce_late()
            self.sync.wait()
        self.opt.step(grad_scale=self.sync.average_scale() if self.sync is not None else 1.0)
        return loss

    def _phase_a(self):
        self.opt.zero_grad(state=self.state)
        if self.cut is not None:
            self.cut.reset()
        sed, doa = self.model(self.x)
        loss = seld_loss_fn(sed, doa, self.target, self.n_sed, *self.w)
        H.backward_from_loss(loss)
        H.join_side_stream()
        return loss.detach()

    def _phase_b(self):
        if self.cut is not None:
            self.cut.finish()
            H.join_side_stream()

    def _phase_c(self):
        self.opt.step(grad_scale=self.sync.average_scale() if self.sync is not None else 1.0, state=self.state)

    def __call__(self, x=None, target=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self._push_lr()
        self.sync_from_host()
        self.graphs[0].replay()
        if self.cut is not None:
            self.sync.reduce_main()
            self.graphs[1].replay()
            self.sync.reduce_late()
            self.sync.wait()
            self.graphs[2].replay()
        self.replays += 1
        self.opt.step_count += 1
        self._step_seen = self.opt.step_count
        # the replayed Adam step rewrote the weights behind the host cache's back: an eager forward after this (validation,
        # an instrumented step) must re-pack the fast-product weight forms
        H.hcq_weights.weights_changed()
        return self.loss

    def sync_from_host(self):
        """Bring the device-resident step state up to date with what happened on the host between replays: eager steps
        or `optimizer.load_state_dict` moved the step count; eager dropout draws moved the host Philox offset (a replay's
        own draws are recorded with offsets from 0 on top of the device base, which the step itself advances)."""
        if self.opt.step_count != self._step_seen:
            self.state[1] = self.opt.step_count
            self._step_seen = self.opt.step_count
        if H.philox.offset:
            self.state[0] += int(H.philox.offset)
            H.philox.offset = 0


_ADAM_DEFAULTS = None


def _torch_adam_defaults():
    """Hyper-parameter keys of this torch version's Adam param group (amsgrad, foreach, capturable, ...)."""
    global _ADAM_DEFAULTS
    if _ADAM_DEFAULTS is None:
        g = torch.optim.Adam([torch.zeros(1, requires_grad=True)]).param_groups[0]
        _ADAM_DEFAULTS = {k: v for k, v in g.items() if k != "params"}
    return _ADAM_DEFAULTS


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) for FlatAdam (train.py:505-508, 570-571)."""

    def __init__(self, optimizer, step_size, gamma=0.1):
        self.opt, self.step_size, self.gamma = optimizer, step_size, gamma
        self.base_lr = optimizer.param_groups[0].setdefault("initial_lr", optimizer.param_groups[0]["lr"])   # as torch does
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.opt.param_groups[0]["lr"] = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)

    def state_dict(self):
        """torch.optim.lr_scheduler.StepLR's keys (train.py:37 saves `scheduler.state_dict()`)."""
        return dict(step_size=self.step_size, gamma=self.gamma, base_lrs=[self.base_lr], last_epoch=self.last_epoch,
                    _step_count=self.last_epoch + 1, _get_lr_called_within_step=False,
                    _last_lr=[self.opt.param_groups[0]["lr"]])

    def load_state_dict(self, sd):
        self.step_size, self.gamma, self.last_epoch = sd["step_size"], sd["gamma"], sd["last_epoch"]
        self.base_lr = sd["base_lrs"][0] if "base_lrs" in sd else sd["base_lr"]


# ------------------------------------------------------------------------------------------
# configuration (train.py:718-838, utility_functions.py:77-91)
# ------------------------------------------------------------------------------------------
def readFile(path):
    """Tokenise a flag file: split on '=' and newlines, 'True' -> '1', 'False' -> 0, drop '#' tokens."""
    with open(path, 'r') as f:
        toks = f.read().replace('=', '+').replace('\n', '+').split('+')
    out = []
    for t in toks:
        if t == 'True':
            out.append('1')
        elif t == 'False':
            out.append(0)
        elif t != '' and '#' not in t:
            out.append(t)
    return out


_FLAGS = [
    ('results_path', str, 'RESULTS/Task2'), ('checkpoint_dir', str, 'RESULTS/Task2'), ('load_model', str, None),
    ('training_predictors_path', str, '/var/datasets/L3DAS21/processed/task2_predictors_train.pkl'),
    ('training_target_path', str, '/var/datasets/L3DAS21/processed/task2_target_train.pkl'),
    ('validation_predictors_path', str, '/var/datasets/L3DAS21/processed/task2_predictors_validation.pkl'),
    ('validation_target_path', str, '/var/datasets/L3DAS21/processed/task2_target_validation.pkl'),
    ('test_predictors_path', str, '/var/datasets/L3DAS21/processed/task2_predictors_test.pkl'),
    ('test_target_path', str, '/var/datasets/L3DAS21/processed/task2_target_test.pkl'),
    ('gpu_id', int, 0), ('use_cuda', str, 'True'), ('early_stopping', str, 'True'), ('fixed_seed', str, 'True'),
    ('lr', float, 0.0001), ('batch_size', int, 1), ('sr', int, 32000), ('patience', int, 250),
    ('architecture', str, 'DualQSELD-TCN'), ('input_channels', int, 4), ('n_mics', int, 1), ('phase', str, 'False'),
    ('class_overlaps', int, 3), ('time_dim', int, 4800), ('freq_dim', int, 256), ('output_classes', int, 14),
    ('pool_size', str, '[[8,2],[8,2],[2,2],[1,1]]'), ('cnn_filters', str, '[64,64,64]'), ('pool_time', str, 'True'),
    ('dropout_perc', float, 0.3), ('D', str, '[10]'), ('G', int, 128), ('U', int, 128), ('V', str, '[128,128]'),
    ('spatial_dropout_rate', float, 0.5), ('batch_norm', str, 'BN'), ('dilation_mode', str, 'fibonacci'),
    ('model_extra_name', str, ''), ('test_mode', str, 'test_best'), ('use_lr_scheduler', str, 'True'),
    ('lr_scheduler_step_size', int, 150), ('lr_scheduler_gamma', float, 0.5), ('min_lr', float, 0.000005),
    ('dataset_normalization', str, 'True'), ('kernel_size_cnn_blocks', int, 3), ('kernel_size_dilated_conv', int, 3),
    ('use_tcn', str, 'True'), ('use_bias_conv', str, 'True'), ('use_bias_linear', str, 'True'), ('verbose', str, 'False'),
    ('sed_loss_weight', float, 1.), ('doa_loss_weight', float, 5.), ('domain_classifier', str, 'same'),
    ('domain', str, 'DQ'), ('fc_activations', str, 'Linear'), ('fc_dropout', str, 'Last'), ('fc_layers', str, '[128]'),
    ('V_kernel_size', int, 3), ('use_time_distributed', str, 'False'), ('parallel_ConvTC_block', str, 'False'),
    ('max_loc_value', float, 2.), ('num_frames', int, 600), ('spatial_threshold', float, 2.),
    ('checkpoint_step', int, 100), ('test_step', int, 10), ('min_n_epochs', int, 1000),
    ('Dcase21_metrics_DOA_threshold', int, 20), ('parallel_magphase', str, 'False'),
    ('TextArgs', str, 'config/Test.txt'),
]
_EVAL = ('use_cuda', 'early_stopping', 'fixed_seed', 'pool_size', 'cnn_filters', 'verbose', 'D', 'V', 'use_lr_scheduler',
         'phase', 'use_tcn', 'use_bias_conv', 'use_bias_linear', 'fc_layers', 'parallel_magphase')
# extensions of this implementation (not in the reference): synthetic data so the step can run without L3DAS21
_EXTRA = [('synthetic', int, 0), ('max_steps', int, 0), ('epochs', int, 0)]


def build_parser():
    p = argparse.ArgumentParser()
    for name, typ, default in _FLAGS + _EXTRA:
        p.add_argument('--' + name, type=typ, default=default)
    return p


def parse_args(argv=None):
    """Two-pass parse like train.py:819-820: argv selects --TextArgs, the file supplies the flags.  Flags the
    reference's parser would reject (e.g. --phm_n in the shipped Q config) are reported, not fatal."""
    parser = build_parser()
    first, _ = parser.parse_known_args(argv)
    tokens = [str(t) for t in readFile(first.TextArgs)] if os.path.isfile(first.TextArgs) else []
    args, unknown = parser.parse_known_args(tokens + list(argv or []))
    if unknown:
        print("ignoring flags unknown to the reference's parser:", unknown, file=sys.stderr)
    for k in _EVAL:
        v = getattr(args, k)
        if isinstance(v, str):
            setattr(args, k, ast.literal_eval(v))
    return args


def model_from_args(args):
    """train.py:448-465."""
    return SELD_Model(time_dim=args.time_dim, freq_dim=args.freq_dim, input_channels=args.input_channels,
                      output_classes=args.output_classes, domain=args.domain, domain_classifier=args.domain_classifier,
                      cnn_filters=args.cnn_filters, kernel_size_cnn_blocks=args.kernel_size_cnn_blocks,
                      pool_size=args.pool_size, pool_time=args.pool_time, D=args.D, dilation_mode=args.dilation_mode,
                      G=args.G, U=args.U, kernel_size_dilated_conv=args.kernel_size_dilated_conv,
                      spatial_dropout_rate=args.spatial_dropout_rate, V=args.V, V_kernel_size=args.V_kernel_size,
                      fc_layers=args.fc_layers, fc_activations=args.fc_activations, fc_dropout=args.fc_dropout,
                      dropout_perc=args.dropout_perc, class_overlaps=args.class_overlaps, use_bias_conv=args.use_bias_conv,
                      use_bias_linear=args.use_bias_linear, batch_norm=args.batch_norm,
                      parallel_ConvTC_block=args.parallel_ConvTC_block, parallel_magphase=args.parallel_magphase,
                      extra_name=args.model_extra_name, verbose=args.verbose)


# ------------------------------------------------------------------------------------------
# checkpoints (train.py:26-81): same dictionary keys, `module.` prefixes stripped on load
# ------------------------------------------------------------------------------------------
def save_model(model, optimizer, state, path, scheduler=None):
    """train.py:26-45: same keys, same value layouts (state dicts of the model / Adam / StepLR, the loop state, the
    three RNG states), so either side reads the other's files."""
    sd = model.module.state_dict() if hasattr(model, "module") else model.state_dict()
    if len(os.path.dirname(path)) > 0 and not os.path.exists(os.path.dirname(path)):
        os.makedirs(os.path.dirname(path))
    ck = {'model_state_dict': sd, 'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else None,
          'state': state}
    if scheduler is not None:
        ck['scheduler_state_dict'] = scheduler.state_dict()
    # the reference's three RNG states (train.py:40-43, read back by index at :77-80) + a fourth entry it never looks at:
    # the position of the dropout kernels' counter-based RNG stream, so that a resumed run does not replay masks
    ck['random_states'] = (np.random.get_state(), torch.get_rng_state(),
                           torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
                           {'seld_philox_offset': H.philox.get_offset()})
    torch.save(ck, path)


def load_model(model, optimizer, path, cuda, device, scheduler=None):
    """train.py:47-81: `module.` prefixes stripped, optimizer / scheduler restored when given, checkpoints that only
    hold `step` accepted, the numpy / torch / device RNG states restored."""
    ck = torch.load(path, map_location=device if cuda else 'cpu', weights_only=False)
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in ck['model_state_dict'].items()}
    target = model.module if hasattr(model, "module") else model
    target.load_state_dict(sd)
    H.hcq_weights.weights_changed()          # the packed weight forms of the fast-product convolutions are stale now
    if optimizer is not None:
        optimizer.load_state_dict(ck['optimizer_state_dict'])
    if scheduler is not None:
        scheduler.load_state_dict(ck['scheduler_state_dict'])
    state = ck['state'] if 'state' in ck else {'step': ck['step']}
    rs = ck.get('random_states')
    if rs is not None:
        np.random.set_state(rs[0])
        torch.set_rng_state(rs[1].cpu())
        if torch.cuda.is_available() and rs[2] is not None:
            torch.cuda.set_rng_state(rs[2].cpu())
    extra = rs[3] if rs is not None and len(rs) > 3 and isinstance(rs[3], dict) else {}
    H.philox.set_offset(int(extra.get('seld_philox_offset', 0)))     # 0 for checkpoints written by the reference
    return state


class CheckpointRotation:
    """The file rotation of the reference's epoch loop (train.py:577-616, 671-684), same file names:

      <model_dir>/checkpoint                              every epoch
      <model_dir>/checkpoint_best_model                   when the validation loss improves
      <model_dir>/checkpoint_best_model_of_checkpoint     the previous best, kept when a new best arrives, or the best
                                                          epoch since that is not the overall best
      <model_dir>checkpoint_epoch_<E>/...                 every `checkpoint_step` epochs: copies of the above
                                                          (the reference concatenates without a separator: kept)

    `end_of_epoch` returns True when the epoch set a new best.  The test-set leg of the rotation
    (`checkpoint_best_model_on_Test`, train.py:623-668) belongs to the metrics row; files that do not
    exist are skipped by the periodic copy where the reference would stop with FileNotFoundError."""

    def __init__(self, model_dir, model_name, checkpoint_step, start_epoch=0):
        self.model_dir, self.model_name, self.checkpoint_step = model_dir, model_name, int(checkpoint_step)
        self.checkpoint_path = os.path.join(model_dir, "checkpoint")
        self.best_path = os.path.join(model_dir, "checkpoint_best_model")
        self.best_of_checkpoint_path = os.path.join(model_dir, "checkpoint_best_model_of_checkpoint")
        self.best_loss_checkpoint = np.inf
        self.best_epoch_checkpoint = start_epoch
        self.new_best = False

    def end_of_epoch(self, model, optimizer, scheduler, state, epoch, val_loss, extra_files=(), periodic_copy=True):
        """Validation-loss bookkeeping and the per-epoch saves (train.py:588-616).  `periodic_copy=False` leaves the
        checkpoint_epoch_<E>/ copies to an explicit `periodic_copy()` call after the test leg (what `main` does)."""
        improved = not (val_loss >= state["best_loss"])
        if not improved:
            state["worse_epochs"] += 1
        else:
            if self.new_best:
                self.best_loss_checkpoint = state["best_loss"]
                self.best_epoch_checkpoint = state["best_epoch"]
                shutil.copyfile(self.best_path, self.best_of_checkpoint_path)
            state["worse_epochs"] = 0
            state["best_loss"] = val_loss
            state["best_epoch"] = epoch
            state["best_checkpoint"] = self.best_path
            self.new_best = True
            save_model(model, optimizer, state, self.best_path, scheduler)
        if val_loss < self.best_loss_checkpoint and (val_loss != state["best_loss"] or self.best_loss_checkpoint == np.inf):
            self.best_loss_checkpoint = val_loss
            save_model(model, optimizer, state, self.best_of_checkpoint_path, scheduler)
            self.best_epoch_checkpoint = epoch
        save_model(model, optimizer, state, self.checkpoint_path, scheduler)
        if periodic_copy:
            self.periodic_copy(state, epoch, extra_files)
        return improved

    def periodic_copy(self, state, epoch, extra_files=()):
        """train.py:671-684: every `checkpoint_step` epochs copy the rotation's files into checkpoint_epoch_<E>/.  The
        reference does this AFTER the test leg of the same epoch (train.py:623-670), so the copied
        checkpoint_best_model_on_Test and the `best_test_epoch` in its name are this epoch's."""
        if self.checkpoint_step > 0 and epoch % self.checkpoint_step == 0:
            d = self.model_dir + 'checkpoint_epoch_{}/'.format(epoch)
            os.makedirs(d, exist_ok=True)
            copies = [(self.best_path, "checkpoint_best_epoch_{}".format(state["best_epoch"])),
                      (self.checkpoint_path, "checkpoint_epoch_{}".format(epoch)),
                      (self.checkpoint_path + '_best_model_on_Test',
                       "checkpoint_best_model_on_Test_epoch_{}".format(state.get("best_test_epoch", 0))),
                      (self.best_of_checkpoint_path,
                       "checkpoint_best_model_checkpoint_epoch_{}".format(self.best_epoch_checkpoint))]
            copies += [(f, self.model_name + "_" + os.path.basename(f)) for f in extra_files]
            for src, name in copies:
                if os.path.isfile(src):
                    shutil.copyfile(src, d + name)

    def test_checkpoint(self, test_mode):
        """train.py:626-642: which checkpoint the test leg evaluates and the epoch it reports: the best model when this
        epoch set a new best, else the best-of-checkpoint model ('test_best' mode); the live model otherwise (None)."""
        if test_mode != 'test_best':
            return None, None
        if self.new_best:
            return self.best_path, None                 # epoch = state['best_epoch'] after loading
        return self.best_of_checkpoint_path, self.best_epoch_checkpoint

    def after_test(self, model, optimizer, scheduler, state, epoch, test_results, test_mode):
        """train.py:655-670: keep `checkpoint_best_model_on_Test` when the Global SELD score (entry 10 of the results)
        is not worse than the best seen, then clear the 'new best' flag.  Returns True when it saved."""
        if not hasattr(self, "best_test_metric"):
            self.best_test_metric = 1
        saved = False
        if test_results[10] <= self.best_test_metric:
            self.best_test_metric = test_results[10]
            if test_mode == 'test_best':
                state["best_test_epoch"] = state["best_epoch"] if self.new_best else self.best_epoch_checkpoint
            else:
                state["best_test_epoch"] = epoch
            save_model(model, optimizer, state, self.checkpoint_path + '_best_model_on_Test', scheduler)
            saved = True
        self.new_best = False
        return saved


# ------------------------------------------------------------------------------------------
# data
# ------------------------------------------------------------------------------------------
def synthetic_batch(batch, channels, freq, time, n_out, seed, device):
    """x ~ N(0,1); target = [Bernoulli(0.1) (B, T/8, n_sed) | U(-1,1) (B, T/8, 3 n_sed)]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, channels, freq, time, generator=g)
    sed = (torch.rand(batch, time // 8, n_out, generator=g) < 0.1).float()
    doa = torch.rand(batch, time // 8, 3 * n_out, generator=g) * 2 - 1
    return x.to(device), torch.cat((sed, doa), 2).to(device)


def load_pickled(pred_path, target_path):
    with open(pred_path, 'rb') as f:
        x = np.array(pickle.load(f))
    with open(target_path, 'rb') as f:
        y = np.array(pickle.load(f))
    return torch.tensor(x).float(), torch.tensor(y).float()


_NORM_OFF = {'False', 'false', 'None', 'none'}
_NORM_UNIT = {'DQ_Normalization', 'UnitNormNormalization', 'UnitNorm'}
_DOMAIN_DQ = ['DQ', 'dq', 'dQ', 'Dual_Quaternion', 'dual_quaternion']


def normalize_dataset(args, *predictors):
    """Dataset normalisation of train.py:242-408, in place on device-resident predictor arrays
    (items, channels, F, T) - the reference's training / validation / test predictors, each normalised with its
    OWN statistics as the reference does.  Returns the description the reference appends to `dataset_string`.

      off ('False' / 'None')                        -> untouched
      'UnitNorm' family, 2 mics, DQ domain          -> unit dual quaternion per position (train.py:257-308);
                                                       with --phase the reference raises ValueError (train.py:310)
      'UnitNorm' family otherwise                   -> untouched (the reference has no branch for it)
      anything else ('True')                        -> (g - mean) / std over the magnitude channels
                                                       (first 4 for 1 mic, first 8 for 2 mics) and, with --phase,
                                                       separately over the remaining channels (train.py:341-405)
    """
    mode = str(args.dataset_normalization)
    if mode in _NORM_OFF:
        return ''
    if mode in _NORM_UNIT:
        if args.n_mics == 2 and args.domain in _DOMAIN_DQ:
            if args.phase:
                raise ValueError('DATASET NORMALIZATION FOR PHASE DUAL QUATERNION NOT YET IMPLEMENTED')
            for x in predictors:
                H.dq_unit_norm_(x)
            return ' Dataset Normalization for 2Mic 8Ch Magnitude Dual Quaternion UnitNorm'
        return ''
    if args.n_mics not in (1, 2):
        return ''
    mag = 4 * args.n_mics
    for x in predictors:
        H.group_standardize_(x, 0, mag)
        if args.phase:
            H.group_standardize_(x, mag, x.shape[1])
    what = f'{args.n_mics}Mic {mag}Ch Magnitude'
    desc = ' Dataset Normalization for ' + what
    if args.phase:
        desc += f' Dataset Normalization for {args.n_mics}Mic {2 * mag}Ch Magnitude-Phase'
    return desc


def test_results_from_counters(counts, total_de, epoch=0):
    """The 16 numbers of train.py:131-150 (with compute_seld_scores, Dcase21_metrics.py:33-49) from the counters."""
    eps_f, eps = sys.float_info.epsilon, np.finfo(float).eps
    TP, FP, FN = counts["TP"], counts["FP"], counts["FN"]
    precision = TP / (TP + FP + eps_f)
    recall = TP / (TP + FN + eps_f)
    F_score = 2 * ((precision * recall) / (precision + recall + eps_f))
    Nref, Nsys = TP + FN, TP + FP
    ER_score = (max(Nref, Nsys) - TP) / (Nref + 0.0)          # ZeroDivisionError without a reference event, as the reference
    ER_dcase21 = (counts["dc_S"] + counts["dc_D"] + counts["dc_I"]) / float(counts["dc_Nref"] + eps)
    F_dcase21 = counts["dc_TP"] / (eps + counts["dc_TP"] + 0.5 * (counts["dc_FP"] + counts["dc_FN"]))
    LE_dcase21 = total_de / float(counts["dc_DE_TP"] + eps) if counts["dc_DE_TP"] else 180
    LR_dcase21 = counts["dc_DE_TP"] / (eps + counts["dc_DE_TP"] + counts["dc_DE_FN"])
    SELD_dcase21 = np.mean([ER_dcase21, 1 - F_dcase21, LE_dcase21 / 180, 1 - LR_dcase21])
    SELD_L3DAS21_LRLE = np.mean([ER_score, 1 - F_score, LE_dcase21 / 180, 1 - LR_dcase21])
    CSL_score = np.mean([LE_dcase21 / 180, 1 - LR_dcase21])
    LSD_score = np.mean([1 - F_score, ER_score])
    return [epoch, F_score, ER_score, precision, recall, TP, FP, FN, CSL_score, LSD_score, SELD_L3DAS21_LRLE, SELD_dcase21,
            ER_dcase21, F_dcase21, LE_dcase21, LR_dcase21]


def evaluate_test(model, device, dataloader, epoch=0, max_loc_value=2., num_frames=600, spatial_threshold=2., args=None):
    """train.py:84-166: same arguments, same 16-entry result list.  The network outputs never leave the device: each
    batch's decode + counting is one kernel (hip_ops.metrics_accumulate) and the counters are read back once."""
    output_classes = args.output_classes if args is not None else 14
    class_overlaps = args.class_overlaps if args is not None else 3
    doa_threshold = args.Dcase21_metrics_DOA_threshold if args is not None else 20
    model.eval()
    acc = H.metrics_new(device)
    with torch.no_grad():
        for x, target in dataloader:
            sed, doa = model(x.to(device))
            # gen_submission_list_task2 is called with its default num_classes = 14 (train.py:110-116)
            H.metrics_accumulate(acc, sed, doa, target.to(device), num_frames, 14, class_overlaps, max_loc_value,
                                 spatial_threshold, doa_threshold)
    counts = dict(zip(H.METRIC_COUNTERS, acc[0].cpu().tolist()))
    results = test_results_from_counters(counts, float(acc[1].item()), epoch)
    print('*******************************\nRESULTS')
    for label, v in (('TP: ', results[5]), ('FP: ', results[6]), ('FN: ', results[7]), ('Global SELD score: ', results[10]),
                     ('LSD score: ', results[9]), ('CSL score: ', results[8]), ('F score: ', results[1]),
                     ('ER score: ', results[2]), ('LE: ', results[14]), ('LR: ', results[15])):
        print(label, v)
    return results


def evaluate(model, device, criterion_sed, criterion_doa, loader, args):
    """Mean loss over a loader, no grad (train.py:168-183)."""
    model.eval()
    total, n = 0.0, 0
    with torch.no_grad():
        for x, target in loader:
            loss = seld_loss(x.to(device), target.to(device), model, criterion_sed, criterion_doa, args)
            n += 1
            total += (loss.item() - total) / n
    return total


def main(args):
    device = torch.device('cuda:' + str(args.gpu_id)) if args.use_cuda else None
    if device is None or not torch.cuda.is_available():
        raise RuntimeError("this implementation has no CPU path: a HIP device is required")
    if args.fixed_seed:
        np.random.seed(1)
        torch.manual_seed(1)
    model = model_from_args(args).to(device)
    print('Total paramters: ' + str(sum(int(np.prod(p.size())) for p in model.parameters())))
    criterion_sed, criterion_doa = BCELoss(), MSELoss()
    optimizer = FlatAdam(model.parameters(), lr=args.lr)
    scheduler = StepLR(optimizer, args.lr_scheduler_step_size, args.lr_scheduler_gamma) if args.use_lr_scheduler else None
    n_out = int(args.output_classes * args.class_overlaps)

    if args.synthetic:
        data = [synthetic_batch(args.batch_size, args.input_channels, args.freq_dim, args.time_dim, n_out, 1234 + i, device)
                for i in range(args.synthetic)]
        tr_data, val_data = data, data[:1]
        test_data = None
    else:
        xs, ys = load_pickled(args.training_predictors_path, args.training_target_path)
        xv, yv = load_pickled(args.validation_predictors_path, args.validation_target_path)
        xs, xv = xs.to(device), xv.to(device)           # 288 GB of HBM: the arrays stay resident, loaders index them
        print(normalize_dataset(args, xs, xv))
        tr_data = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xs, ys), args.batch_size, shuffle=True,
                                              pin_memory=False)
        val_data = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xv, yv), args.batch_size, shuffle=False,
                                               pin_memory=False)
        test_data = None
        if os.path.isfile(str(args.test_predictors_path)) and os.path.isfile(str(args.test_target_path)):
            xt, yt = load_pickled(args.test_predictors_path, args.test_target_path)
            xt = xt.to(device)
            print(normalize_dataset(args, xt))
            test_data = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xt, yt), 1, shuffle=False)

    model_dir = os.path.join(args.checkpoint_dir, model.model_name)
    os.makedirs(model_dir, exist_ok=True)
    state = {"step": 0, "worse_epochs": 0, "epochs": 0, "best_loss": np.inf, "best_epoch": 0, "best_test_epoch": 0}
    epoch = 0
    if args.load_model is not None and os.path.isfile(args.load_model):
        print("Continuing training full model from checkpoint " + str(args.load_model))
        state = load_model(model, optimizer, args.load_model, args.use_cuda, device, scheduler)
        epoch = state["epochs"]
    rotation = CheckpointRotation(model_dir, model.model_name, args.checkpoint_step, start_epoch=epoch)
    while (state["worse_epochs"] < args.patience or epoch < args.min_n_epochs) and not (args.epochs and epoch >= args.epochs):
        epoch += 1
        state["epochs"] += 1
        model.train()
        train_loss = 0.0
        t0 = time.time()
        for i, (x, target) in enumerate(tr_data):
            optimizer.zero_grad()
            loss = seld_loss(x.to(device), target.to(device), model, criterion_sed, criterion_doa, args)
            loss.backward()
            optimizer.step()
            state["step"] += 1
            train_loss += (loss.detach() - train_loss) / (i + 1)
            if args.max_steps and state["step"] >= args.max_steps:
                break
        val_loss = evaluate(model, device, criterion_sed, criterion_doa, val_data, args)
        if scheduler is not None and optimizer.param_groups[0]['lr'] > args.min_lr:
            scheduler.step()
        print(f"epoch {epoch}: train {float(train_loss):.5f} val {val_loss:.5f} lr {optimizer.param_groups[0]['lr']:.2e} "
              f"({time.time() - t0:.1f}s)")
        if rotation.end_of_epoch(model, optimizer, scheduler, state, epoch, val_loss, periodic_copy=False):
            print("MODEL IMPROVED ON VALIDATION SET!")
        if test_data is not None and args.test_step > 0 and epoch % args.test_step == 0:      # train.py:623-670
            path, at_epoch = rotation.test_checkpoint(args.test_mode)
            if path is not None:
                live = os.path.join(model_dir, "checkpoint")
                state = load_model(model, optimizer, path, args.use_cuda, device, scheduler)
                at_epoch = state['best_epoch'] if at_epoch is None else at_epoch
            results = evaluate_test(model, device, test_data, epoch=epoch if path is None else at_epoch,
                                    max_loc_value=args.max_loc_value, num_frames=args.num_frames,
                                    spatial_threshold=args.spatial_threshold, args=args)
            rotation.after_test(model, optimizer, scheduler, state, epoch, results, args.test_mode)
            if path is not None:
                # the reference reloads args.load_model here (train.py:667), which only exists when resuming; the live
                # checkpoint of this epoch is what training must continue from
                best_test_epoch = state.get("best_test_epoch", 0)
                state = load_model(model, optimizer, live, args.use_cuda, device, scheduler)
                state["best_test_epoch"] = best_test_epoch      # set by after_test on the state loaded for the test
        rotation.periodic_copy(state, epoch)                    # after the test leg, as train.py:671-684
        if args.max_steps and state["step"] >= args.max_steps:
            break
    return state


if __name__ == '__main__':
    args = parse_args(sys.argv[1:])
    main(args)
