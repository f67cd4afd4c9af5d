"""Conditional predictor-corrector sampling (super-resolution, inpainting, edge->photo).

Mirrors sampling/conditional.py of the reference: ``get_conditional_sampling_fn`` (:8-45) with
the same ``'default'`` sentinels, and ``get_pc_conditional_sampler`` (:47-228) returning
``pc_conditional_sampler(model, y, show_evolution=False) -> (x, info)``.

Fast path: for the (conditional_reverse_diffusion, conditional_langevin, VE) pair - what every
BASELINE config selects - the whole loop runs on the device through csd_pc_sample
(sampling/fused.py).  Any other registered predictor/corrector pair runs the reference's
per-step protocol (corrector then predictor, sampling/conditional.py:208-211) by calling the
objects' ``update_fn``; ``use_path=True`` runs the bridge sampler of :124-178.  Extra keyword arguments (not in the reference): ``noise_tape`` (list of
standard-normal tensors in the reference's draw order, SURVEY.md 3.1 - parity mode) and ``seed``
(on-device Philox - throughput mode).
"""
import functools

import torch

from ..models import utils as mutils
from . import fused
from .correctors import NoneCorrector, get_corrector
from .predictors import NonePredictor, get_predictor


def get_conditional_sampling_fn(config, sde, shape, eps, predictor='default', corrector='default',
                                p_steps='default', c_steps='default', snr='default', denoise='default',
                                use_path='default'):
    predictor = get_predictor((config.sampling.predictor if predictor == 'default' else predictor).lower())
    corrector = get_corrector((config.sampling.corrector if corrector == 'default' else corrector).lower())
    if p_steps == 'default':
        p_steps = config.model.num_scales
    if c_steps == 'default':
        c_steps = config.sampling.n_steps_each
    if snr == 'default':
        snr = config.sampling.snr
    if denoise == 'default':
        denoise = config.sampling.noise_removal
    if use_path == 'default':
        use_path = False
    return get_pc_conditional_sampler(sde=sde, shape=shape, predictor=predictor, corrector=corrector, snr=snr,
                                      p_steps=p_steps, c_steps=c_steps,
                                      probability_flow=config.sampling.probability_flow,
                                      continuous=config.training.continuous, denoise=denoise,
                                      use_path=use_path, eps=eps)


def conditional_shared_predictor_update_fn(x, y, t, sde, model, predictor, probability_flow, continuous):
    score_fn = mutils.get_conditional_score_fn(
        mutils.get_score_fn(sde, model, conditional=True, train=False, continuous=continuous), 'x')
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    obj = (NonePredictor if predictor is None else predictor)(c_sde, score_fn, probability_flow)
    return obj.update_fn(x, y, t)


def conditional_shared_corrector_update_fn(x, y, t, sde, model, corrector, continuous, snr, n_steps):
    score_fn = mutils.get_conditional_score_fn(
        mutils.get_score_fn(sde, model, conditional=True, train=False, continuous=continuous), 'x')
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    obj = (NoneCorrector if corrector is None else corrector)(c_sde, score_fn, snr, n_steps)
    return obj.update_fn(x, y, t)


def get_pc_conditional_sampler(sde, shape, predictor, corrector, snr, p_steps, c_steps=1, probability_flow=False,
                               continuous=False, denoise=True, use_path=False, eps=1e-5):
    two_sde = isinstance(sde, dict) and len(sde) == 2
    if use_path and not two_sde:
        raise NotImplementedError('use_path needs the two-SDE (CMDE / VS-CMDE) setting: sde = {"x": ..., "y": ...}')
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    pred_fn = functools.partial(conditional_shared_predictor_update_fn, sde=sde, predictor=predictor,
                                probability_flow=probability_flow, continuous=continuous)
    corr_fn = functools.partial(conditional_shared_corrector_update_fn, sde=sde, corrector=corrector,
                                continuous=continuous, snr=snr, n_steps=c_steps)

    def perturbed(update_fn, x, y, t, model):
        """one conditional update; CMDE/VS-CMDE draw a fresh y_t first (sampling/conditional.py:104-116)"""
        with torch.no_grad():
            vec_t = torch.ones(x.shape[0], device=model.device) * t
            if two_sde:
                from .. import ops
                std = sde['y'].marginal_prob(y, vec_t)[1]
                y_in = y + ops.scale_rows(torch.randn_like(y), std)
            else:
                y_in = y
            x, x_mean = update_fn(x=x, y=y_in, t=vec_t, model=model)
        return x, x_mean, y_in

    def path_sampler(model, y, show_evolution=False):
        """``use_path`` (sampling/conditional.py:85-100,124-178): y_t follows the bridge p(y_t | y_0, y_{t+tau})
        (sde_lib.py:323-339) instead of being redrawn from the marginal; predictor first, then the corrector on the
        same y_t.  The bridge coefficients are host scalars, the pixel work runs on csd_axpby."""
        from .. import ops
        sy = sde['y']

        def scalars(fn):          # evaluate a [B]-valued sde function at one time on the host, in fp32 like the reference
            return [float(v.flatten()[0]) for v in fn()]

        with torch.no_grad():
            x = c_sde.prior_sampling(shape).to(model.device)
            timesteps = torch.linspace(c_sde.T, eps, p_steps)
            tau, T0 = timesteps[0] - timesteps[1], timesteps[0]
            std0, = scalars(lambda: (sy.marginal_prob(torch.zeros(1, 1, 1, 1), (T0 + tau).reshape(1))[1],))
            y_tpt = ops.axpby(y, torch.randn_like(y), 1.0, std0)
            evolution = {'x': [], 'y': []}
            x_mean = x
            one, zero = torch.ones(1, 1, 1, 1), torch.zeros(1, 1, 1, 1)
            for i in range(p_steps):
                t1 = timesteps[i].reshape(1)
                w0, std = scalars(lambda: sy.compute_backward_kernel(one, zero, t1, tau.reshape(1)))
                w1, = scalars(lambda: (sy.compute_backward_kernel(zero, one, t1, tau.reshape(1))[0],))
                vec_t = torch.ones(x.shape[0], device=model.device) * float(timesteps[i])
                y_t = ops.axpby(ops.axpby(y, y_tpt, w0, w1), torch.randn_like(y), 1.0, std)
                x, x_mean = pred_fn(x=x, y=y_t, t=vec_t, model=model)
                y_tpt = y_t
                x, x_mean = corr_fn(x=x, y=y_tpt, t=vec_t, model=model)
                if show_evolution:
                    evolution['x'].append(x.cpu())
                    evolution['y'].append(y_tpt.cpu())
            out = x_mean if denoise else x
            if show_evolution:
                return out, {'evolution': {'x': torch.stack(evolution['x']), 'y': torch.stack(evolution['y'])}}
            return out, {}

    if use_path:
        def pc_path_sampler(model, y, show_evolution=False, noise_tape=None, seed=None, global_norm=None):
            """the bridge sampler on the fused device loop (csd_pc_params.path_coef) when the pair is fusable and no y_t evolution is
            asked for; else step by step (path_sampler above)"""
            if global_norm is not None:
                raise NotImplementedError('use_path is not provided in the global-norm sharded mode (sample_sharded(global_norm=True))')
            if not show_evolution and fused.fusable(model, sde, predictor, corrector, c_steps, probability_flow, continuous, use_path=True):
                x, _, _ = fused.run(model, sde, shape, y, p_steps, snr, eps, denoise, noise_tape=noise_tape, seed=seed,
                                    predictor=predictor, corrector=corrector, probability_flow=probability_flow, use_path=True)
                return x, {}
            if noise_tape is not None:
                raise NotImplementedError('noise_tape is only available on the fused path')
            return path_sampler(model, y, show_evolution)
        return pc_path_sampler

    def pc_conditional_sampler(model, y, show_evolution=False, noise_tape=None, seed=None, global_norm=None):
        if fused.fusable(model, sde, predictor, corrector, c_steps, probability_flow, continuous):
            x, rec, _ = fused.run(model, sde, shape, y, p_steps, snr, eps, denoise, noise_tape=noise_tape,
                                  seed=seed, record=show_evolution, global_norm=global_norm, predictor=predictor, corrector=corrector,
                                  probability_flow=probability_flow)
            if show_evolution:
                return x, {'evolution': {'x': rec.cpu(), 'y': None}}
            return x, {}
        if noise_tape is not None:
            raise NotImplementedError('noise_tape is only available on the fused path')
        if global_norm is not None:
            # the step-by-step fallback (c_steps != 1, a VP SDE, a model that is not a HipUNet) computes per-shard norms: refusing is
            # better than silently not delivering "identical to one process"; use the 'conditional_langevin_global' corrector there
            raise NotImplementedError('global-norm sharded sampling runs on the fused device loop only; this (model, sde, predictor, '
                                      'corrector, c_steps) combination falls back to the step-by-step loop')
        with torch.no_grad():
            x = c_sde.prior_sampling(shape).to(model.device)
            evolution = {'x': [], 'y': []}
            timesteps = torch.linspace(c_sde.T, eps, p_steps, device=model.device)
            x_mean = x
            for i in range(p_steps):
                t = timesteps[i]
                x, x_mean, y_p = perturbed(corr_fn, x, y, t, model)
                x, x_mean, y_p = perturbed(pred_fn, x, y, t, model)
                if show_evolution:
                    evolution['x'].append(x.cpu())
                    evolution['y'].append(y_p.cpu())
            out = x_mean if denoise else x
            if show_evolution:
                return out, {'evolution': {'x': torch.stack(evolution['x']), 'y': torch.stack(evolution['y'])}}
            return out, {}

    return pc_conditional_sampler
