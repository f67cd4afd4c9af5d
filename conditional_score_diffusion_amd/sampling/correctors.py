"""Corrector registry + the Langevin corrector on the HIP update kernels.

Mirrors sampling/correctors.py of the reference: ``register_corrector`` / ``get_corrector``
(:5-27), ``Corrector`` (:29-49), ``langevin`` (:51-78), ``conditional_langevin`` (:81-108),
``none`` / ``conditional_none`` (:145-163).  ``ald`` / ``conditional_ald`` (:111-142) are not
provided yet and raise NotImplementedError.
"""
import abc

import torch

from .. import ops, sde_lib

_CORRECTORS = {}


def register_corrector(cls=None, *, name=None):
    def _register(c):
        key = c.__name__ if name is None else name
        if key in _CORRECTORS:
            raise ValueError(f'Already registered model with name: {key}')
        _CORRECTORS[key] = c
        return c

    return _register if cls is None else _register(cls)


def get_corrector(name):
    return _CORRECTORS[name]


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde = sde
        self.score_fn = score_fn
        self.snr = snr
        self.n_steps = n_steps

    @abc.abstractmethod
    def update_fn(self, x, t):
        """-> (x, x_mean)"""


def _langevin(sde, score_of, x, snr, n_steps):
    """n_steps Langevin updates with the batch-mean step size (sampling/correctors.py:100-106)."""
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.cVPSDE, sde_lib.subVPSDE)):
        raise NotImplementedError('the HIP Langevin step covers the VE SDEs (alpha = 1); got %s'
                                  % sde.__class__.__name__)
    x = x.clone()
    x_mean = x
    for _ in range(n_steps):
        grad = score_of(x)
        noise = torch.randn_like(x)
        x, x_mean = ops.langevin_step(x, grad, noise, 1.0, snr)
    return x, x_mean


@register_corrector(name='langevin')
class LangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        return _langevin(self.sde, lambda v: self.score_fn(v, t), x, self.snr, self.n_steps)


@register_corrector(name='conditional_langevin')
class conditionalLangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.cVESDE, sde_lib.cVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, y, t):
        return _langevin(self.sde, lambda v: self.score_fn(v, y, t), x, self.snr, self.n_steps)


@register_corrector(name='none')
class NoneCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        pass

    def update_fn(self, x, t):
        return x, x


@register_corrector(name='conditional_none')
class conditionalNoneCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        pass

    def update_fn(self, x, y, t):
        return x, x


def _not_yet(name):
    class _Missing(Corrector):
        def __init__(self, *a, **k):
            raise NotImplementedError('corrector %r is not provided by the HIP path yet (SURVEY.md 8f)' % name)

        def update_fn(self, x, t):  # pragma: no cover
            raise NotImplementedError

    _Missing.__name__ = 'Missing_' + name
    return _Missing


for _n in ('ald', 'conditional_ald'):
    register_corrector(_not_yet(_n), name=_n)
