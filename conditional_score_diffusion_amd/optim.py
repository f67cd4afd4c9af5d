"""Optimizer, warm-up / clipping manager and EMA of the training step, on flat fp32 buffers and ONE fused HIP kernel.

Mirrors the reference's ``get_optimizer`` / ``optimization_manager`` (losses.py:26-53: torch.optim.Adam with betas
(beta1, 0.999), linear warm-up, ``clip_grad_norm_``) and ``ExponentialMovingAverage`` (models/ema.py:16-140,
decay min(decay, (1+n)/(10+n))), re-designed for the MI355X: all parameters live in one contiguous buffer (the
``nn.Parameter`` objects become views of it), gradients accumulate into one contiguous buffer (a single RCCL
all-reduce per bucket, one norm kernel), and ``step()`` is a single HBM-bound pass - csd_adam_step: clip + Adam + EMA,
5 reads + 4 writes per element - instead of ~10 elementwise launches per parameter tensor.
"""
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr


# Parameter -> the live FlatParams that holds its storage.  Kept OUTSIDE the Parameter (keyed by id, validated by identity): an
# attribute on the Parameter would be serialised by Parameter.__reduce_ex__ and a weakref there breaks torch.save(model),
# copy.deepcopy(optimizer) and module pickling for multiprocessing spawn.
_OWNERS = {}


def _owner(p):
    """the live FlatParams that holds this Parameter's storage, or None"""
    ent = _OWNERS.get(id(p))
    if ent is None:
        return None
    pref, fref = ent
    if pref() is not p or fref() is None:          # (the id was recycled, or the flat buffer is gone)
        _OWNERS.pop(id(p), None)
        return None
    return fref()


class FlatParams:
    """Moves ``params`` into one contiguous fp32 buffer (each parameter becomes a view) with a matching gradient buffer.

    A parameter can live in ONE flat buffer only.  ``FlatParams.of(params)`` is what the optimizer and the EMA call: it hands
    back the existing FlatParams when exactly these parameters were flattened before - so ``get_optimizer(config,
    model.parameters())`` and ``ExponentialMovingAverage(model.parameters(), decay)``, the two calls the reference makes
    (lightning_modules/BaseSdeGenerativeModel.py:75-96), share one buffer - and raises when the sets only overlap."""

    @classmethod
    def of(cls, params):
        if isinstance(params, FlatParams):
            return params
        plist = [p for p in params if p.requires_grad]
        owners = {id(o): o for o in map(_owner, plist) if o is not None}
        if not owners:
            return cls(plist)
        if len(owners) == 1:
            flat = next(iter(owners.values()))
            if len(flat.params) == len(plist) and all(a is b for a, b in zip(flat.params, plist)):
                return flat
        raise RuntimeError('FlatParams: some of these parameters already live in another flat buffer (a different parameter '
                           'list was flattened before); build the optimizer and the EMA from the same parameter list')

    # ---- serialisation helpers: optimizer / EMA state is stored PER PARAMETER (the reference's layout: torch.optim.Adam state and
    # models/ema.py shadow_params are per-parameter lists), never as the raw flat buffer - its padding is an internal alignment
    # choice that must not leak into checkpoints (round-5 advisor finding)
    def split(self, buf):
        """flat buffer (this layout) -> list of per-parameter tensors (copies, parameter shapes, parameter order)"""
        return [buf[int(o):int(o) + p.numel()].detach().clone().view_as(p) for p, o in zip(self.params, self.offsets[:-1])]

    def merge_into(self, buf, value):
        """the inverse of split(); also accepts the two raw flat forms older checkpoints hold: this layout (padded, numel equal) or
        the unpadded concatenation of the parameters (written before the 16-byte alignment)"""
        if isinstance(value, torch.Tensor):
            if value.numel() == self.numel:
                buf.copy_(value.reshape(-1))
                return
            if value.numel() != sum(p.numel() for p in self.params):
                raise ValueError('flat state of %d elements matches neither this layout (%d) nor the unpadded one (%d)'
                                 % (value.numel(), self.numel, sum(p.numel() for p in self.params)))
            flat, at, value = value.reshape(-1), 0, []
            for p in self.params:
                value.append(flat[at:at + p.numel()])
                at += p.numel()
        if len(value) != len(self.params):
            raise ValueError('state holds %d tensors for %d parameters' % (len(value), len(self.params)))
        buf.zero_()
        for p, o, v in zip(self.params, self.offsets[:-1], value):
            if v.numel() != p.numel():
                raise ValueError('state tensor of %d elements for a parameter of %d' % (v.numel(), p.numel()))
            buf[int(o):int(o) + p.numel()].copy_(v.reshape(-1))

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        if any(_owner(p) is not None for p in self.params):
            raise RuntimeError('FlatParams: parameters are already flattened - use FlatParams.of(params)')
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise RuntimeError('FlatParams: parameters must be float32 on one device')
        # every parameter starts on a 16-byte boundary (4 floats): the planned backward writes the gradient views directly and the
        # library's vector stores need the alignment - a 6-float bias (the output pyramids of NCSN++ on 6 channels) would otherwise
        # misalign everything behind it; the padding holds zeros for ever (zero gradient -> zero Adam update)
        starts, o = [], 0
        for p in self.params:
            starts.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.offsets = np.asarray(starts + [o], dtype=np.int64)
        self.numel = int(o)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets[:-1]):
            o = int(o)
            self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)      # autograd accumulates in place into the flat buffer
            _OWNERS[id(p)] = (weakref.ref(p), weakref.ref(self))
        _lib.WEIGHT_EPOCH[0] += 1

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets[:-1]):        # (re-attach: a set_to_none zero_grad elsewhere detaches the views)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * int(o):
                p.grad = self.grad[int(o):int(o) + p.numel()].view_as(p)


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (losses.py:12-23) on a FlatParams; ``step()`` = csd_global_norm + csd_adam_step.
    A ``torch.optim.Optimizer`` (one parameter group), so the reference's ``LambdaLR`` warm-up scheduler
    (BaseSdeGenerativeModel.py:86-96) accepts it.  One difference from torch.optim.Adam: a parameter that received no
    gradient in a step is updated with a zero gradient (moments decay) instead of being skipped."""

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = FlatParams.of(params if isinstance(params, FlatParams) else list(params))
        _lib.require_gpu_tensor(self.flat.data, 'parameters')          # the update is a HIP kernel: no CPU fallback
        super().__init__(self.flat.params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.num_steps = 0
        self.max_norm = -1.0          # set by optimize_fn (grad_clip); < 0: no clipping
        self.last_grad_norm = None    # device scalar of the last step (no host sync)
        self._norm_scratch = None

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def grad_norm(self):
        out = torch.empty(1, dtype=torch.float32, device=self.flat.grad.device)
        if self._norm_scratch is None:
            self._norm_scratch = torch.empty(lib().csd_global_norm_scratch_bytes(), dtype=torch.uint8, device=out.device)
        check(lib().csd_global_norm(ptr(self.flat.grad), ptr(out), self.flat.numel, ptr(self._norm_scratch),
                                    current_stream(out.device)), 'global_norm')
        return out

    def step(self, closure=None, ema=None):
        """One update; ``ema`` (an ExponentialMovingAverage over the same FlatParams) is folded into the same pass."""
        if closure is not None:
            raise NotImplementedError('FusedAdam.step: closures are not supported')
        g = self.param_groups[0]
        self.num_steps += 1
        norm = self.grad_norm() if self.max_norm >= 0 else None
        self.last_grad_norm = norm
        decay, shadow = 0.0, None
        if ema is not None:
            if ema.flat is not self.flat:
                raise ValueError('the EMA must track the optimizer\'s FlatParams to be fused into the step')
            decay, shadow = ema._next_decay(), ema.shadow
        check(lib().csd_adam_step(ptr(self.flat.data), ptr(self.flat.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(shadow),
                                  ptr(norm), self.flat.numel, self.num_steps, float(g['lr']), float(g['betas'][0]),
                                  float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), float(self.max_norm),
                                  float(decay), current_stream(self.flat.data.device)), 'adam_step')
        _lib.WEIGHT_EPOCH[0] += 1

    def state_dict(self):
        return {'format': 2, 'num_steps': self.num_steps, 'exp_avg': self.flat.split(self.exp_avg), 'exp_avg_sq': self.flat.split(self.exp_avg_sq),
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        self.num_steps = int(sd['num_steps'])
        self.flat.merge_into(self.exp_avg, sd['exp_avg'])           # per-parameter lists (format 2) or a raw flat buffer (older checkpoints)
        self.flat.merge_into(self.exp_avg_sq, sd['exp_avg_sq'])
        self.param_groups[0].update(sd['param_groups'][0])


def get_optimizer(config, params):
    """losses.py:12-23."""
    if config.optim.optimizer != 'Adam':
        raise NotImplementedError(f'Optimizer {config.optim.optimizer} not supported yet!')
    return FusedAdam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                     weight_decay=config.optim.weight_decay)


def optimization_manager(config):
    """losses.py:26-53: returns ``optimize_fn(optimizer, params, step, ...)`` - warm-up, clipping (negative disables), step."""

    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup, grad_clip=config.optim.grad_clip,
                    ema=None):
        if warmup > 0:
            for g in optimizer.param_groups:
                g['lr'] = float(lr * np.minimum(step / warmup, 1.0))
        optimizer.max_norm = float(grad_clip)
        optimizer.step(ema=ema)

    return optimize_fn


class ExponentialMovingAverage:
    """models/ema.py:16-140 on one flat shadow buffer (update = csd_ema_update, or fused into FusedAdam.step)."""

    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.flat = FlatParams.of(parameters if isinstance(parameters, FlatParams) else list(parameters))
        _lib.require_gpu_tensor(self.flat.data, 'parameters')
        self.shadow = self.flat.data.clone()
        self._stored = None

    def _next_decay(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return decay

    def update(self, parameters=None):
        check(lib().csd_ema_update(ptr(self.shadow), ptr(self.flat.data), self.flat.numel, float(self._next_decay()),
                                   current_stream(self.shadow.device)), 'ema_update')

    def copy_to(self, parameters=None):
        self.flat.data.copy_(self.shadow)
        _lib.WEIGHT_EPOCH[0] += 1

    def store(self, parameters=None):
        self._stored = self.flat.data.clone()

    def restore(self, parameters=None):
        if self._stored is None:
            raise RuntimeError('This ExponentialMovingAverage has no `store()`ed weights to `restore()`')
        self.flat.data.copy_(self._stored)
        self._stored = None
        _lib.WEIGHT_EPOCH[0] += 1

    def state_dict(self):
        # the reference's keys (models/ema.py:86-88): decay, num_updates, shadow_params (a per-parameter list)
        return {'decay': self.decay, 'num_updates': self.num_updates, 'shadow_params': self.flat.split(self.shadow)}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = sd['decay'], sd['num_updates']
        self.flat.merge_into(self.shadow, sd['shadow_params'] if 'shadow_params' in sd else sd['shadow'])
