"""Unconditional predictor-corrector sampling.

Mirrors ``get_sampling_fn`` (sampling/unconditional.py:13-75) and ``get_pc_sampler`` (:161-228):
``pc_sampler(model, show_evolution=False) -> (samples, {'times', 'steps'[, 'evolution']})``.
``sampling.method='ode'`` gives the probability-flow ODE sampler (:93-158): scipy's black-box RK45 on the host, every drift
evaluation = one network evaluation + one HIP axpby.  ``get_inpainting_fn`` / ``get_pc_inpainter`` (:78-91, :230-345): the PC loop
with the known pixels re-imposed after every update (masked blend on csd_axpby / csd_mul).
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
        return get_ode_sampler(sde=sde, shape=shape, denoise=denoise, eps=eps)
    if method != 'pc':
        raise ValueError(f"Sampler name {config.sampling.method} unknown.")
    return get_pc_sampler(sde=sde, shape=shape, predictor=predictor, corrector=corrector, snr=snr,
                          p_steps=p_steps, c_steps=c_steps, probability_flow=config.sampling.probability_flow,
                          continuous=config.training.continuous, denoise=denoise, eps=eps)


def get_ode_sampler(sde, shape, denoise=False, rtol=1e-5, atol=1e-5, method='RK45', eps=1e-3):
    """Probability-flow ODE sampler with a black-box solver (sampling/unconditional.py:93-158):
    ``ode_sampler(model, z=None) -> (samples, nfe)``.  The drift of the reverse-time ODE, f(x, t) - g(t)^2 score / 2
    (sde_lib.py:123-133), is linear in x for every SDE of sde_lib, so an evaluation is the score network plus one
    ``csd_axpby`` with host scalars; the state crosses to the host per evaluation exactly as in the reference."""
    from scipy import integrate

    from .. import ops
    from .predictors import ReverseDiffusionPredictor, _linear_sde_coeffs

    def denoise_update_fn(model, x):
        score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
        predictor_obj = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False)
        vec_eps = torch.ones(x.shape[0], device=x.device) * eps
        _, x = predictor_obj.update_fn(x, vec_eps)
        return x

    def drift_fn(model, x, t):
        score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
        phi, g = _linear_sde_coeffs(sde, t)
        return ops.axpby(x, score_fn(x, t), alpha=phi, beta=-0.5 * g * g)

    def ode_sampler(model, z=None):
        with torch.no_grad():
            x = sde.prior_sampling(shape).to(model.device) if z is None else z

            def ode_func(t, xf):
                xt = mutils.from_flattened_numpy(xf, shape).to(model.device).type(torch.float32)
                vec_t = torch.ones(shape[0], device=xt.device) * t
                return mutils.to_flattened_numpy(drift_fn(model, xt, vec_t))

            solution = integrate.solve_ivp(ode_func, (sde.T, eps), mutils.to_flattened_numpy(x), rtol=rtol, atol=atol,
                                           method=method)
            nfe = solution.nfev
            x = torch.tensor(solution.y[:, -1]).reshape(shape).to(model.device).type(torch.float32)
            if denoise:
                x = denoise_update_fn(model, x)
            return x, nfe

    return ode_sampler


def get_inpainting_fn(config, sde, eps, n_steps_each=1):
    """sampling/unconditional.py:78-91."""
    return get_pc_inpainter(sde=sde, predictor=get_predictor(config.sampling.predictor.lower()),
                            corrector=get_corrector(config.sampling.corrector.lower()), snr=config.sampling.snr,
                            n_steps=n_steps_each, probability_flow=config.sampling.probability_flow,
                            continuous=config.training.continuous, denoise=config.sampling.noise_removal, eps=eps)


def get_pc_inpainter(sde, predictor, corrector, snr, n_steps=1, probability_flow=False, continuous=False, denoise=True, eps=1e-5):
    """Image inpainting with an unconditional model (sampling/unconditional.py:230-345):
    ``pc_inpainter(model, data, mask, show_evolution=False) -> (x, info)``; ``mask`` is 1 on known pixels.  After every
    corrector / predictor update the known region is replaced by the data perturbed to the current noise level."""
    from .. import ops
    from ..losses import _bstd

    pred_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor, probability_flow=probability_flow,
                                continuous=continuous)
    corr_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector, continuous=continuous, snr=snr,
                                n_steps=n_steps)

    def blend(a, b, mask):
        """a*(1 - mask) + b*mask = a + (b - a)*mask on the device"""
        from ..grad_ops import _mul
        return ops.axpby(a, _mul(ops.axpby(b, a, 1.0, -1.0), mask))

    def inpaint_update(update_fn, model, data, mask, x, t):
        vec_t = torch.ones(data.shape[0], device=data.device) * t
        x, x_mean = update_fn(x, vec_t, model=model)
        m, std = _bstd(sde, data, vec_t)                       # mean scale / std of p_t(x | data): [B] host scalars
        mean = data if bool(torch.all(m == 1)) else ops.scale_rows(data, m.to(data.device))
        masked = ops.axpby(mean, ops.scale_rows(torch.randn_like(x), std.to(data.device)))
        x = blend(x, masked, mask)
        x_mean = blend(x, mean, mask)
        return x, x_mean

    def pc_inpainter(model, data, mask, show_evolution=False):
        with torch.no_grad():
            data, mask = data.contiguous().float(), mask.contiguous().float()
            x = blend(sde.prior_sampling(data.shape).to(data.device), data, mask)
            evolution = [x.cpu()] if show_evolution else None
            timesteps = torch.linspace(sde.T, eps, sde.N)
            x_mean = x
            for i in range(sde.N):
                t = timesteps[i]
                x, x_mean = inpaint_update(corr_fn, model, data, mask, x, t)
                x, x_mean = inpaint_update(pred_fn, model, data, mask, x, t)
                if show_evolution:
                    evolution.append(x.cpu())
            info = {'evolution': torch.stack(evolution)} if show_evolution else {}
            return (x_mean if denoise else x), info

    return pc_inpainter


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

    def pc_sampler(model, show_evolution=False, noise_tape=None, seed=None, global_norm=None):
        steps = p_steps * (c_steps + 1)
        if fused.fusable(model, sde, predictor, corrector, c_steps, probability_flow, continuous):
            label = 'fourier' if getattr(model, 'embedding_type', 'positional') == 'fourier' else 'sigma'
            x, rec, ts = fused.run(model, sde, shape, None, p_steps, snr, eps, denoise, noise_tape=noise_tape,
                                   seed=seed, record=show_evolution, unconditional_label=label, global_norm=global_norm,
                                   predictor=predictor, corrector=corrector, probability_flow=probability_flow)
            info = {'times': ts, 'steps': steps}
            if show_evolution:
                info['evolution'] = rec.cpu()
            return x, info
        if noise_tape is not None:
            raise NotImplementedError('noise_tape is only available on the fused path')
        if global_norm is not None:       # (the step-by-step fallback would use per-shard norms: refuse instead; 'langevin_global' exists)
            raise NotImplementedError('global-norm sharded sampling runs on the fused device loop only; this (model, sde, predictor, '
                                      'corrector, c_steps) combination falls back to the step-by-step loop')
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
