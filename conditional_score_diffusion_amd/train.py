"""One data-parallel training step of the score network: loss -> backward -> gradient all-reduce -> clip + Adam + EMA.

Replaces ``BaseSdeGenerativeModel.training_step`` / ``configure_optimizers`` (lightning_modules/BaseSdeGenerativeModel.py:
52-96), the EMA callback (lightning_modules/callbacks.py:119-133, models/ema.py) and Lightning-DDP's gradient all-reduce
(run_lib.py:55-73) with one process per GPU over ``torch.distributed`` (RCCL):

* forward + backward on the HIP operators of grad_ops (csrc/backward.hip);
* gradients accumulate into ONE flat buffer, all-reduced in 32 MiB buckets that are launched from autograd hooks while the
  backward of earlier layers is still running (distributed.GradSync);
* ONE fused kernel applies clipping + Adam + EMA (optim.FusedAdam / csd_adam_step).
"""
import torch

from . import losses, optim
from .distributed import GradSync


class Trainer:
    def __init__(self, config, model, sde, group=None, bucket_bytes=32 << 20):
        self.config, self.model, self.sde = config, model, sde
        t = config.training
        conditional = isinstance(sde, dict) or model.__class__.__name__ != 'DDPM' and getattr(model, 'y_channels', 0) > 0
        self.loss_fn = losses.get_general_sde_loss_fn(sde, True, conditional=conditional, reduce_mean=t.reduce_mean,
                                                      continuous=t.continuous, likelihood_weighting=t.likelihood_weighting)
        self.eval_loss_fn = losses.get_general_sde_loss_fn(sde, False, conditional=conditional, reduce_mean=t.reduce_mean,
                                                           continuous=t.continuous, likelihood_weighting=t.likelihood_weighting)
        self.flat = optim.FlatParams(model.parameters())
        self.optimizer = optim.get_optimizer(config, self.flat)
        self.optimize_fn = optim.optimization_manager(config)
        self.ema = optim.ExponentialMovingAverage(self.flat, decay=config.model.ema_rate)
        self.sync = GradSync(self.flat, group, bucket_bytes)
        self.step = 0                     # completed optimizer steps (the warm-up factor of step k is k / warmup)

    def train_step(self, batch):
        """-> detached loss of this rank's shard.  ``batch`` is the loss_fn's: ``x`` or ``(y, x)``."""
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model, batch)
        self.sync.scale_loss(loss).backward()
        self.sync.finish()
        # Lightning's gradient_clip_val (run_lib.py:58-59): 0 disables; losses.optimization_manager: negative disables
        clip = float(self.config.optim.grad_clip)
        self.optimize_fn(self.optimizer, self.flat.params, self.step, grad_clip=clip if clip > 0 else -1.0, ema=self.ema)
        self.step += 1
        return loss.detach()

    # -- checkpoint / resume (Lightning keeps the same pieces in its .ckpt: state_dict, optimizer_states, the EMA callback state) --
    def state_dict(self):
        return {'model': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'optimizer': {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()},
                'ema': {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.ema.state_dict().items()},
                'step': self.step, 'train_calls': getattr(self.model, '_train_calls', 0)}

    def load_state_dict(self, sd):
        """Resume: parameters, Adam moments, EMA shadow, step counter and the dropout stream position - the next train_step is
        bit-identical to the one the saved run would have taken."""
        from . import _lib
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(sd['model'][k])                       # (in place: the parameters stay views of the flat buffer)
        _lib.WEIGHT_EPOCH[0] += 1
        self.optimizer.load_state_dict(sd['optimizer'])
        self.ema.load_state_dict(sd['ema'])
        self.step = int(sd['step'])
        if hasattr(self.model, '_train_calls'):
            self.model._train_calls = int(sd['train_calls'])

    @torch.no_grad()
    def eval_loss(self, batch, use_ema=False):
        if use_ema:
            self.ema.store()
            self.ema.copy_to()
        try:
            return self.eval_loss_fn(self.model, batch)
        finally:
            if use_ema:
                self.ema.restore()
