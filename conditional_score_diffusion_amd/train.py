"""One data-parallel training step of the score network: loss -> backward -> gradient all-reduce -> clip + Adam + EMA.

Replaces ``BaseSdeGenerativeModel.training_step`` / ``configure_optimizers`` (lightning_modules/BaseSdeGenerativeModel.py:
52-96), the EMA callback (lightning_modules/callbacks.py:119-133, models/ema.py) and Lightning-DDP's gradient all-reduce
(run_lib.py:55-73) with one process per GPU over ``torch.distributed`` (RCCL):

* forward + backward as ONE planned graph behind csd_unet_train_forward / csd_unet_backward (DDPM family; csrc/train_graph.h) or
  on the differentiable HIP operators of grad_ops (NCSN++; csrc/backward.hip);
* gradients live in ONE flat buffer, all-reduced in 32 MiB buckets (distributed.GradSync).  On the operator-granular executor the
  buckets are launched from autograd hooks while the backward of earlier layers is still running.  On the planned graph (the whole
  backward is ONE csd_unet_backward call that only enqueues kernels) the library records one gradient-ready event per bucket on
  the backward's stream (csd_unet_backward_marks); after the call returns - the GPU is still early in the backward - every bucket
  is launched on a communication stream that waits for its own event only.  (Measured on ONE GPU only: the event order and the
  gradients are tested, the overlap itself has never run on more than one device.)
* ONE fused kernel applies clipping + Adam + EMA (optim.FusedAdam / csd_adam_step).
"""
import torch

from . import losses, optim
from .distributed import GradSync


def get_reduction_fn(y0, xk, yk):
    """lightning_callbacks/callbacks.py:81-86: starts at y0 and reaches yk after xk steps at an inverse multiplicative rate."""
    def f(x):
        return xk * yk * y0 / (x * (y0 - yk) + xk * yk)
    return f


class Trainer:
    def __init__(self, config, model, sde, group=None, bucket_bytes=32 << 20):
        self.config, self.model, self.sde = config, model, sde
        # VS-CMDE (DecreasingVarianceConfigurationSetterCallback, lightning_callbacks/callbacks.py:23-78): the conditioning SDE's
        # sigma_max_y / sigma_min_y shrink with the global step; the SDE object of 'y' is rebuilt before every training batch
        m = config.model
        has = (lambda k: k in m) if hasattr(m, '__contains__') else (lambda k: hasattr(m, k))
        self._vs = None
        if isinstance(sde, dict) and has('reach_target_steps') and has('sigma_max_y_target'):
            self._vs = (get_reduction_fn(m.sigma_max_y, m.reach_target_steps, m.sigma_max_y_target),
                        get_reduction_fn(m.sigma_min_y, m.reach_target_steps, m.sigma_min_y_target if has('sigma_min_y_target') else m.sigma_min_y))
        self.sigma_max_y = float(m.sigma_max_y) if has('sigma_max_y') else None
        self.sigma_min_y = float(m.sigma_min_y) if has('sigma_min_y') else None
        self._build_loss_fns()
        self.flat = optim.FlatParams.of(model.parameters())
        self.optimizer = optim.get_optimizer(config, self.flat)
        self.optimize_fn = optim.optimization_manager(config)
        self.ema = optim.ExponentialMovingAverage(self.flat, decay=config.model.ema_rate)
        self.sync = GradSync(self.flat, group, bucket_bytes)
        # planned training graph (csd_unet_backward): the gradients are written straight into the flat buffer's views (p.grad, zeroed
        # by zero_grad()) - autograd has nothing to accumulate and GradSync.finish() reduces every bucket after the backward
        if getattr(model, 'train_executor', None) == 'planned':
            model.grad_sink = True
            self.sync.attach_planned(model)           # per-bucket gradient-ready events: the all-reduce overlaps the backward
        self.step = 0                     # completed optimizer steps (the warm-up factor of step k is k / warmup)

    def close(self):
        """retire the gradient synchronisation: autograd hooks removed, gradient-ready events destroyed, the network handle's marks
        erased if they are still this Trainer's (a Trainer rebuilt on the same model registers its own and keeps them)"""
        self.sync.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _build_loss_fns(self):
        config, model, sde = self.config, self.model, self.sde
        t = config.training
        conditional = isinstance(sde, dict) or model.__class__.__name__ != 'DDPM' and getattr(model, 'y_channels', 0) > 0
        self.loss_fn = losses.get_general_sde_loss_fn(sde, True, conditional=conditional, reduce_mean=t.reduce_mean,
                                                      continuous=t.continuous, likelihood_weighting=t.likelihood_weighting)
        self.eval_loss_fn = losses.get_general_sde_loss_fn(sde, False, conditional=conditional, reduce_mean=t.reduce_mean,
                                                           continuous=t.continuous, likelihood_weighting=t.likelihood_weighting)

    def reconfigure_conditioning_sde(self):
        """callbacks.py:44-56 + ConditionalSdeGenerativeModel.reconfigure_conditioning_sde (:179-195): sde['y'] at the current step"""
        from . import sde_lib
        self.sigma_max_y, self.sigma_min_y = float(self._vs[0](self.step)), float(self._vs[1](self.step))
        self.sde['y'] = sde_lib.VESDE(sigma_min=self.sigma_min_y, sigma_max=self.sigma_max_y, N=self.config.model.num_scales)
        self._build_loss_fns()

    def train_step(self, batch, global_n=None):
        """-> detached loss of this rank's shard.  ``batch`` is the loss_fn's: ``x`` or ``(y, x)``.  ``global_n``: images of the GLOBAL
        batch when the shards are ragged (a global batch that does not divide by the world size, distributed.shard_bounds): the
        rank's mean loss is then weighted by its share of the images instead of 1 / world."""
        if self._vs is not None:
            self.reconfigure_conditioning_sde()
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model, batch)
        if global_n:
            first = batch[0] if isinstance(batch, (tuple, list)) else batch
            self.sync.scale_loss(loss, local_n=first.shape[0], global_n=global_n).backward()
        else:
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
                'step': self.step, 'train_calls': getattr(self.model, '_train_calls', 0),
                'sigma_max_y': self.sigma_max_y, 'sigma_min_y': self.sigma_min_y}     # (the buffers the Lightning module registers)

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
