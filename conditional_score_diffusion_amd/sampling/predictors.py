"""Predictor registry + the reverse-diffusion predictor on the HIP update kernel.

Mirrors sampling/predictors.py of the reference: ``register_predictor`` / ``get_predictor``
(:6-28), the ``Predictor`` base (:30-50) and the classes registered under
``reverse_diffusion`` (:79-89), ``conditional_reverse_diffusion`` (:92-102) and ``none`` /
``conditional_none`` (:182-200).  ``update_fn`` keeps the reference signature
``(x, t) | (x, y, t) -> (x, x_mean)``; the pixel update runs in csrc/sampler.hip.

Not yet provided: euler_maruyama, ancestral_sampling and their
conditional variants - requesting them raises NotImplementedError instead of silently running
somewhere else.
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


def _not_yet(name):
    class _Missing(Predictor):
        def __init__(self, *a, **k):
            raise NotImplementedError('predictor %r is not provided by the HIP path yet (SURVEY.md 8f)' % name)

        def update_fn(self, x, t):  # pragma: no cover
            raise NotImplementedError

    _Missing.__name__ = 'Missing_' + name
    return _Missing


for _n in ('euler_maruyama', 'conditional_euler_maruyama', 'ancestral_sampling', 'conditional_ancestral_sampling'):
    register_predictor(_not_yet(_n), name=_n)
