"""Predictor registry + the reverse-diffusion predictor on the HIP update kernel.

Mirrors sampling/predictors.py of the reference: ``register_predictor`` / ``get_predictor``
(:6-28), the ``Predictor`` base (:30-50) and the classes registered under
``reverse_diffusion`` (:79-89), ``conditional_reverse_diffusion`` (:92-102) and ``none`` /
``conditional_none`` (:182-200).  ``update_fn`` keeps the reference signature
``(x, t) | (x, y, t) -> (x, x_mean)``; the pixel update runs in csrc/sampler.hip.

``euler_maruyama`` (:52-76) and ``ancestral_sampling`` (:105-179) and their conditional variants run on the
general affine update kernel (``csd_affine_noise_step``): their per-step coefficients are host scalars.
"""
import abc

import torch

from .. import ops, sde_lib

_PREDICTORS = {}


def register_predictor(cls=None, *, name=None):
    def _register(c):
        key = c.__name__ if name is None else name
        if key in _PREDICTORS:
            raise ValueError(f'Already registered model with name: {key}')
        _PREDICTORS[key] = c
        return c

    return _register if cls is None else _register(cls)


def get_predictor(name):
    return _PREDICTORS[name]


class Predictor(abc.ABC):
    """Base class: builds the reverse SDE from the score function."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn, probability_flow)
        self.score_fn = score_fn
        self.probability_flow = probability_flow

    @abc.abstractmethod
    def update_fn(self, x, t):
        """-> (x, x_mean)"""


def _uniform_scalar(v, what):
    """The HIP step kernels take one scalar per call (the sampler uses a constant t per step)."""
    v0 = float(v.flatten()[0])
    if v.numel() > 1 and not bool(torch.all(v == v.flatten()[0])):
        raise NotImplementedError('per-sample %s within one update is not supported by the HIP step kernel' % what)
    return v0


def _ve_reverse_diffusion(sde, score, x, t, probability_flow):
    """x_mean = x + G^2*score (f = 0 for VE); x = x_mean + G*z  (sde_lib.py:135-140,353-362)."""
    if not isinstance(sde, (sde_lib.VESDE, sde_lib.cVESDE)):
        raise NotImplementedError('the HIP reverse-diffusion step covers the VE SDEs; got %s'
                                  % sde.__class__.__name__)
    if probability_flow:
        raise NotImplementedError('probability-flow predictor is not provided by the HIP step kernel yet')
    G = _uniform_scalar(sde.discretize(torch.zeros(t.shape[0], 1), t.detach().cpu())[1], 'G')
    z = torch.randn_like(x)
    return ops.reverse_diffusion_step(x.clone(), score, z, 1.0, G)


@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, t):
        return _ve_reverse_diffusion(self.sde, self.score_fn(x, t), x, t, self.probability_flow)


@register_predictor(name='conditional_reverse_diffusion')
class conditionalReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, y, t):
        return _ve_reverse_diffusion(self.sde, self.score_fn(x, y, t), x, t, self.probability_flow)


@register_predictor(name='none')
class NonePredictor(Predictor):
    """Corrector-only sampling (sampling/predictors.py:182-190)."""

    def __init__(self, sde, score_fn, probability_flow=False):
        pass

    def update_fn(self, x, t):
        return x, x


@register_predictor(name='conditional_none')
class conditionalNonePredictor(Predictor):
    def __init__(self, sde, score_fn, probability_flow=False):
        pass

    def update_fn(self, x, y, t):
        return x, x


def _linear_sde_coeffs(sde, t):
    """(phi, g) with forward drift f(x, t) = phi*x and diffusion g at the (uniform) time t: every SDE of
    sde_lib has a drift that is linear in x, so one evaluation at x = 1 on the host gives phi."""
    t1 = t.detach().cpu().flatten()[:1].to(torch.float32)
    drift, diffusion = sde.sde(torch.ones(1, 1, 1, 1), t1)
    return float(drift.flatten()[0]), float(diffusion.flatten()[0])


def _euler_maruyama(sde, score, x, t, probability_flow):
    """x_mean = x + (f - g^2*score*(1/2 if ode else 1))*dt, x = x_mean + g*sqrt(-dt)*z with dt = -1/N
    (sampling/predictors.py:52-76; reverse drift sde_lib.py:123-133)."""
    _uniform_scalar(t, 't')
    phi, g = _linear_sde_coeffs(sde, t)
    dt = -1.0 / sde.N
    kappa = 0.5 if probability_flow else 1.0
    z = torch.randn_like(x)
    return ops.affine_noise_step(x.clone(), score, z, 1.0 + phi * dt, -kappa * g * g * dt,
                                 0.0 if probability_flow else g * (-dt) ** 0.5)


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, t):
        return _euler_maruyama(self.sde, self.score_fn(x, t), x, t, self.probability_flow)


@register_predictor(name='conditional_euler_maruyama')
class conditionalEulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, y, t):
        return _euler_maruyama(self.sde, self.score_fn(x, y, t), x, t, self.probability_flow)


def _ancestral(sde, score, x, t):
    """sampling/predictors.py:114-135: VE  x_mean = x + (s_i^2 - s_{i-1}^2)*score, std = sqrt(s_{i-1}^2 (s_i^2 - s_{i-1}^2)/s_i^2);
    VP  x_mean = (x + beta_i*score)/sqrt(1 - beta_i), std = sqrt(beta_i)."""
    _uniform_scalar(t, 't')
    t1 = t.detach().cpu().flatten()[:1].to(torch.float32)
    i = int((t1 * (sde.N - 1) / sde.T).long()[0])
    z = torch.randn_like(x)
    if isinstance(sde, (sde_lib.VESDE, sde_lib.cVESDE)):
        sig = sde.discrete_sigmas.to(torch.float32)
        s2 = float(sig[i]) ** 2
        a2 = float(sig[i - 1]) ** 2 if i > 0 else 0.0
        return ops.affine_noise_step(x.clone(), score, z, 1.0, s2 - a2, (a2 * (s2 - a2) / s2) ** 0.5)
    beta = float(sde.discrete_betas.to(torch.float32)[i])
    r = (1.0 - beta) ** 0.5
    return ops.affine_noise_step(x.clone(), score, z, 1.0 / r, beta / r, beta ** 0.5)


@register_predictor(name='ancestral_sampling')
class AncestralSamplingPredictor(Predictor):
    """Ancestral sampling; VE / VP SDEs only, as in the reference (sampling/predictors.py:105-143)."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn, probability_flow)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"

    def update_fn(self, x, t):
        return _ancestral(self.sde, self.score_fn(x, t), x, t)


@register_predictor(name='conditional_ancestral_sampling')
class conditionalAncestralSamplingPredictor(Predictor):
    """The reference registers this class with a two-argument ``update_fn(self, x, t)`` that calls the
    three-argument helpers (sampling/predictors.py:175-179), so it cannot run inside the conditional loop
    there.  Here ``update_fn(x, y, t)`` has the signature the conditional sampler uses."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn, probability_flow)
        if not isinstance(sde, (sde_lib.cVESDE, sde_lib.cVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"

    def update_fn(self, x, y, t):
        return _ancestral(self.sde, self.score_fn(x, y, t), x, t)
