"""Score-matching losses with the reference's factories (losses.py:55-232): evaluation value and training loss.

``get_sde_loss_fn`` / ``get_general_sde_loss_fn`` return ``loss_fn(model, batch) -> scalar tensor`` with the reference's
argument meaning: per-sample ``t ~ U(eps, T)``, perturbation ``x_t = mean + std * z``, score evaluation, and the
(likelihood-)weighted denoising score-matching residual, reduced per sample (mean or 0.5*sum) and averaged over the
batch.  Every pixel-sized operation runs on the HIP kernels (perturbation: csd_scale_rows + csd_axpby; residual:
csd_scale_rows + csd_axpby; per-sample squared norms: csd_row_norms); the remaining arithmetic is on B scalars.

``train=False`` (the reference's eval step) evaluates the network on the planned graph executor and returns a plain value.
``train=True`` puts the model in training mode: the network runs on the differentiable HIP operators (grad_ops: forward and
backward kernels of csrc/backward.hip, dropout on), the residual / per-sample reduction are differentiable HIP operators too,
and the returned scalar is autograd-connected - ``loss.backward()`` fills ``param.grad`` of every network parameter
(SURVEY.md 8 rows a19/a20).  The last reduction over the B per-sample values is torch arithmetic on a [B] tensor.
"""
import torch

from . import grad_ops, ops, sde_lib
from .models import utils as mutils
from .optim import get_optimizer, optimization_manager  # noqa: F401  (losses.py:26-53: same names, same module as the reference)


def _bstd(sde, ref, t):
    """(mean-scale, std) of p_t(x|x0) as [B] fp32 host tensors: every SDE of sde_lib has mean = m(t)*x0."""
    t = t.detach().cpu().float()
    mean, std = sde.marginal_prob(torch.ones(t.shape[0], 1, 1, 1), t)
    return mean.flatten(), std.flatten().float()


def _perturb(x, z, m, std):
    """x_t = m_b * x + std_b * z on the device."""
    dev = x.device
    xm = x if bool(torch.all(m == 1)) else ops.scale_rows(x, m.to(dev))
    return ops.axpby(xm, ops.scale_rows(z, std.to(dev)))


def _residual_sumsq(score, z, std, weighting):
    """per-sample sum of squares of (score*std + z) [weighting False] or (score + z/std) [True] -> [B] host tensor."""
    dev = score.device
    if score.requires_grad:            # training: differentiable operators, [B] device tensor
        if weighting:
            d = grad_ops.axpby(score, ops.scale_rows(z, std.to(dev), divide=True))
        else:
            d = grad_ops.axpby(grad_ops.scale_rows(score, std.to(dev)), z)
        return grad_ops.sumsq_rows(d)
    if weighting:
        d = ops.axpby(score, ops.scale_rows(z, std.to(dev), divide=True))
    else:
        d = ops.axpby(ops.scale_rows(score, std.to(dev)), z)
    n = ops.row_norms(d).cpu().double()
    return n * n


def _reduce(sumsq, numel, reduce_mean):
    return sumsq / numel if reduce_mean else 0.5 * sumsq


def _g2(sde, t):
    t = t.detach().cpu().float()
    return (sde.sde(torch.zeros(t.shape[0], 1, 1, 1), t)[1].flatten().double()) ** 2


def get_sde_loss_fn(sde, train, reduce_mean=True, continuous=True, likelihood_weighting=True, eps=1e-5):
    """losses.py:55-97 (unconditional)."""

    def loss_fn(model, batch):
        score_fn = mutils.get_score_fn(sde, model, train=train, continuous=continuous)
        t = torch.rand(batch.shape[0]) * (sde.T - eps) + eps
        z = torch.randn_like(batch)
        m, std = _bstd(sde, batch, t)
        score = score_fn(_perturb(batch, z, m, std), t.to(batch.device))
        per = batch[0].numel()
        losses = _reduce(_residual_sumsq(score, z, std, likelihood_weighting), per, reduce_mean)
        if likelihood_weighting:
            losses = losses * _g2(sde, t).to(losses)
        return losses.mean().float()

    return loss_fn


def get_general_sde_loss_fn(sde, train, conditional=False, reduce_mean=True, continuous=True, likelihood_weighting=True,
                            eps=1e-5):
    """losses.py:99-232: unconditional, SR3 (one conditional SDE, clean y) and the two-SDE CMDE / VS-CMDE branch."""
    if not conditional:
        return get_sde_loss_fn(sde, train, reduce_mean, continuous, likelihood_weighting, eps)
    if isinstance(sde, dict):
        if len(sde) != 2:
            raise NotImplementedError('multi-speed losses with >= 3 SDEs (losses.py:148-183) are not provided')
        assert likelihood_weighting, 'For the variance reduction technique in inverse problems, we only support likelihood weighting for the time being.'

        def loss_fn(model, batch):
            y, x = batch
            score_fn = mutils.get_score_fn(sde, model, conditional=True, train=train, continuous=continuous)
            t = torch.rand(x.shape[0]) * (sde['x'].T - eps) + eps
            z_y = torch.randn_like(y)
            m_y, std_y = _bstd(sde['y'], y, t)
            z_x = torch.randn_like(x)
            m_x, std_x = _bstd(sde['x'], x, t)
            score = score_fn({'x': _perturb(x, z_x, m_x, std_x), 'y': _perturb(y, z_y, m_y, std_y)}, t.to(x.device))
            sx = _residual_sumsq(score['x'].contiguous(), z_x, std_x, True)
            sx = sx * _g2(sde['x'], t).to(sx)
            sy = _residual_sumsq(score['y'].contiguous(), z_y, std_y, True)
            sy = sy * _g2(sde['y'], t).to(sy)
            numel = x[0].numel() + y[0].numel()          # the reference concatenates both residuals before reducing
            return _reduce(sx + sy, numel, reduce_mean).mean().float()

        return loss_fn

    def loss_fn(model, batch):          # SR3 estimator (losses.py:185-205)
        y, x = batch
        score_fn = mutils.get_score_fn(sde, model, conditional=True, train=train, continuous=continuous)
        t = torch.rand(x.shape[0]) * (sde.T - eps) + eps
        z = torch.randn_like(x)
        m, std = _bstd(sde, x, t)
        score = score_fn({'x': _perturb(x, z, m, std), 'y': y}, t.to(x.device))
        losses = _reduce(_residual_sumsq(score, z, std, likelihood_weighting), x[0].numel(), reduce_mean)
        if likelihood_weighting:
            losses = losses * _g2(sde, t).to(losses)
        return losses.mean().float()

    return loss_fn


# ------------------------------------------------------------------------------------------------------------------
# discrete-time ("legacy") objectives and the one-step function - losses.py:236-407
# ------------------------------------------------------------------------------------------------------------------
def _diff_sumsq(a, b):
    """per-sample sum of squares of (a - b): differentiable in ``a`` when it carries a graph"""
    if a.requires_grad:
        return grad_ops.sumsq_rows(grad_ops.axpby(a, b, 1.0, -1.0))
    n = ops.row_norms(ops.axpby(a, b, 1.0, -1.0)).cpu().double()
    return n * n


def get_smld_loss_fn(vesde, train, reduce_mean=False, likelihood_weighting=False):
    """losses.py:236-265 (SMLD / NCSN objective on the discrete sigma ladder).  (score + z/sigma)^2 sigma^2 = (score sigma + z)^2:
    both weightings of the reference reduce to the same per-sample value."""
    assert isinstance(vesde, sde_lib.VESDE), "SMLD training only works for VESDEs."

    def loss_fn(model, batch):
        score_fn = mutils.get_score_fn(vesde, model, train=train)
        labels = torch.randint(0, vesde.N, (batch.shape[0],))
        sigmas = vesde.discrete_sigmas[labels].float()
        z = torch.randn_like(batch)
        perturbed = ops.axpby(batch, ops.scale_rows(z, sigmas.to(batch.device)))
        score = score_fn(perturbed, (labels / (vesde.N - 1)).to(batch.device))
        losses = _reduce(_residual_sumsq(score, z, sigmas, False), batch[0].numel(), reduce_mean)
        return losses.mean().float()

    return loss_fn


def get_inverse_problem_smld_loss_fn(sde, train, reduce_mean=False, likelihood_weighting=True):
    """losses.py:267-318: SMLD objective for the {'x', 'y'} pair of VE SDEs on a shared label."""
    def loss_fn(model, batch):
        y, x = batch
        score_fn = mutils.get_score_fn(sde, model, train=train)
        labels = torch.randint(0, sde['x'].N, (x.shape[0],))
        sig_y, sig_x = sde['y'].discrete_sigmas[labels].float(), sde['x'].discrete_sigmas[labels].float()
        z_y = torch.randn_like(y)
        z_x = torch.randn_like(x)
        pert = {'x': ops.axpby(x, ops.scale_rows(z_x, sig_x.to(x.device))), 'y': ops.axpby(y, ops.scale_rows(z_y, sig_y.to(y.device)))}
        score = score_fn(pert, (labels / (sde['x'].N - 1)).to(x.device))
        numel = x[0].numel() + y[0].numel()
        if likelihood_weighting:      # (s + z/sigma)^2 sigma^2 per domain, concatenated, reduced
            sx = _residual_sumsq(score['x'].contiguous(), z_x, sig_x, False)
            sy = _residual_sumsq(score['y'].contiguous(), z_y, sig_y, False)
            return _reduce(sx + sy, numel, reduce_mean).mean().float()
        sx = _residual_sumsq(score['x'].contiguous(), z_x, sig_x, True)      # (s + z/sigma)^2
        sy = _residual_sumsq(score['y'].contiguous(), z_y, sig_y, True)
        w = (sig_x.double() ** 2 * sig_y.double() ** 2) / (sig_x.double() ** 2 + sig_y.double() ** 2)
        tot = sx + sy
        return (_reduce(tot, numel, reduce_mean) * w.to(tot)).mean().float()

    return loss_fn


def get_ddpm_loss_fn(vpsde, train, reduce_mean=True):
    """losses.py:320-340 (DDPM noise-prediction objective)."""
    assert isinstance(vpsde, sde_lib.VPSDE), "DDPM training only works for VPSDEs."

    def loss_fn(model, batch):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vpsde.N, (batch.shape[0],))
        a = vpsde.sqrt_alphas_cumprod[labels].float()
        b = vpsde.sqrt_1m_alphas_cumprod[labels].float()
        z = torch.randn_like(batch)
        perturbed = ops.axpby(ops.scale_rows(batch, a.to(batch.device)), ops.scale_rows(z, b.to(batch.device)))
        out = model_fn(perturbed, labels.to(batch.device))
        return _reduce(_diff_sumsq(out, z), batch[0].numel(), reduce_mean).mean().float()

    return loss_fn


def get_inverse_problem_ddpm_loss_fn(vpsdes, train, reduce_mean=True):
    """losses.py:342-343: not implemented upstream either (the reference returns the NotImplementedError class)."""
    raise NotImplementedError('discrete DDPM training of inverse problems is not implemented in the reference')


def get_step_fn(sde, train, optimize_fn=None, reduce_mean=False, continuous=True, likelihood_weighting=False):
    """losses.py:345-407: ``step_fn(state, batch) -> loss`` with state = {'model', 'optimizer', 'ema', 'step'}."""
    if continuous:
        loss_fn = get_sde_loss_fn(sde, train, reduce_mean=reduce_mean, continuous=True, likelihood_weighting=likelihood_weighting)
    elif isinstance(sde, dict):
        if isinstance(sde['y'], sde_lib.VESDE) and isinstance(sde['x'], sde_lib.cVESDE) and len(sde) == 2:
            loss_fn = get_inverse_problem_smld_loss_fn(sde, train, reduce_mean=reduce_mean, likelihood_weighting=likelihood_weighting)
        else:
            raise NotImplementedError('This combination of sdes is not supported for discrete training yet.')
    elif isinstance(sde, sde_lib.VESDE):
        loss_fn = get_smld_loss_fn(sde, train, reduce_mean=reduce_mean)
    elif isinstance(sde, sde_lib.VPSDE):
        loss_fn = get_ddpm_loss_fn(sde, train, reduce_mean=reduce_mean)
    else:
        raise ValueError(f"Discrete training for {sde.__class__.__name__} is not recommended.")

    def step_fn(state, batch):
        model = state['model']
        if train:
            optimizer = state['optimizer']
            optimizer.zero_grad()
            loss = loss_fn(model, batch)
            loss.backward()
            optimize_fn(optimizer, model.parameters(), step=state['step'])
            state['step'] += 1
            state['ema'].update(model.parameters())
        else:
            with torch.no_grad():
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch)
                ema.restore(model.parameters())
        return loss

    return step_fn
