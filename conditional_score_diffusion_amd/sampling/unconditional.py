"""Unconditional predictor-corrector sampling.

Mirrors ``get_sampling_fn`` (sampling/unconditional.py:13-75) and ``get_pc_sampler`` (:161-228):
``pc_sampler(model, show_evolution=False) -> (samples, {'times', 'steps'[, 'evolution']})``.
The ODE sampler (:93-158) and the inpainter (:230-345) are not on any BASELINE config and are not
provided (``sampling.method='ode'`` raises NotImplementedError).
"""
import functools

import torch

from ..models import utils as mutils
from . import fused
from .correctors import NoneCorrector, get_corrector
from .predictors import NonePredictor, get_predictor


def get_sampling_fn(config, sde, shape, eps, predictor='default', corrector='default', p_steps='default',
                    c_steps='default', snr='default', denoise='default'):
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
    method = config.sampling.method.lower()
    if method == 'ode':
        raise NotImplementedError('the probability-flow ODE sampler is not provided by the HIP path')
    if method != 'pc':
        raise ValueError(f"Sampler name {config.sampling.method} unknown.")
    return get_pc_sampler(sde=sde, shape=shape, predictor=predictor, corrector=corrector, snr=snr,
                          p_steps=p_steps, c_steps=c_steps, probability_flow=config.sampling.probability_flow,
                          continuous=config.training.continuous, denoise=denoise, eps=eps)


def shared_predictor_update_fn(x, t, sde, model, predictor, probability_flow, continuous):
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=continuous)
    obj = (NonePredictor if predictor is None else predictor)(sde, score_fn, probability_flow)
    return obj.update_fn(x, t)


def shared_corrector_update_fn(x, t, sde, model, corrector, continuous, snr, n_steps):
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=continuous)
    obj = (NoneCorrector if corrector is None else corrector)(sde, score_fn, snr, n_steps)
    return obj.update_fn(x, t)


def get_pc_sampler(sde, shape, predictor, corrector, snr, p_steps, c_steps, probability_flow=False,
                   continuous=False, denoise=True, eps=1e-3):
    pred_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                probability_flow=probability_flow, continuous=continuous)
    corr_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector, continuous=continuous,
                                snr=snr, n_steps=c_steps)

    def pc_sampler(model, show_evolution=False, noise_tape=None, seed=0):
        steps = p_steps * (c_steps + 1)
        if fused.fusable(model, sde, predictor, corrector, c_steps, probability_flow, continuous):
            label = 'fourier' if getattr(model, 'embedding_type', 'positional') == 'fourier' else 'sigma'
            x, rec, ts = fused.run(model, sde, shape, None, p_steps, snr, eps, denoise, noise_tape=noise_tape,
                                   seed=seed, record=show_evolution, unconditional_label=label)
            info = {'times': ts, 'steps': steps}
            if show_evolution:
                info['evolution'] = rec.cpu()
            return x, info
        if noise_tape is not None:
            raise NotImplementedError('noise_tape is only available on the fused path')
        with torch.no_grad():
            x = sde.prior_sampling(shape).to(model.device).type(torch.float32)
            timesteps = torch.linspace(sde.T, eps, p_steps, device=model.device)
            evolution = []
            x_mean = x
            for i in range(p_steps):
                vec_t = torch.ones(shape[0], device=model.device) * timesteps[i]
                x, x_mean = corr_fn(x, vec_t, model=model)
                x, x_mean = pred_fn(x, vec_t, model=model)
                if show_evolution:
                    evolution.append(x.cpu())
            info = {'times': timesteps, 'steps': steps}
            if show_evolution:
                info['evolution'] = torch.stack(evolution)
            return (x_mean if denoise else x), info

    return pc_sampler
