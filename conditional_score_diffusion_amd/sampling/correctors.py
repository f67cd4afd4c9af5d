"""Corrector registry + the Langevin corrector on the HIP update kernels.

Mirrors sampling/correctors.py of the reference: ``register_corrector`` / ``get_corrector``
(:5-27), ``Corrector`` (:29-49), ``langevin`` (:51-78), ``conditional_langevin`` (:81-108),
``none`` / ``conditional_none`` (:145-163) and ``ald`` (:111-142, plus a conditional variant) on the general
affine update kernel.
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


def _alpha(sde, t):
    """sampling/correctors.py:63-67,94-98: alphas[timestep] for the VP / subVP SDEs, 1 for the VE SDEs.  One host scalar per call:
    the samplers pass the same time for every sample of the batch (sampling/conditional.py:206, unconditional.py:211)."""
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.cVPSDE, sde_lib.subVPSDE)):
        tt = t.detach().float()
        if tt.numel() > 1 and not bool((tt == tt.reshape(-1)[0]).all()):
            # the reference's alpha is per sample (alphas[timestep] of shape [B]); this mirror carries ONE scalar into the kernel
            raise NotImplementedError('Langevin corrector for VP / subVP SDEs: the time must be the same for every sample of the batch')
        timestep = (tt.reshape(-1)[:1].cpu() * (sde.N - 1) / sde.T).long()
        return float(sde.alphas[timestep])
    return 1.0


def _langevin(sde, score_of, x, snr, n_steps, t):
    """n_steps Langevin updates with the batch-mean step size (sampling/correctors.py:69-76,100-106)."""
    alpha = _alpha(sde, t)
    x = x.clone()
    x_mean = x
    for _ in range(n_steps):
        grad = score_of(x)
        noise = torch.randn_like(x)
        x, x_mean = ops.langevin_step(x, grad, noise, 1.0, snr, alpha)
    return x, x_mean


@register_corrector(name='langevin')
class LangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        return _langevin(self.sde, lambda v: self.score_fn(v, t), x, self.snr, self.n_steps, t)


@register_corrector(name='conditional_langevin')
class conditionalLangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.cVESDE, sde_lib.cVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, y, t):
        return _langevin(self.sde, lambda v: self.score_fn(v, y, t), x, self.snr, self.n_steps, t)


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


def _langevin_global(sde, score_of, x, snr, n_steps, t=None, group=None):
    """Langevin corrector whose batch-mean norms run over the GLOBAL batch of a sharded run ("global-norm"
    exactness mode): per-sample norms on the device, ONE all-reduce of two fp32 sums per corrector step, then the
    same update as sampling/correctors.py:100-106.  Without an initialised process group it equals ``langevin``."""
    import torch.distributed as dist
    alpha = _alpha(sde, t) if t is not None else 1.0
    x = x.clone()
    x_mean = x
    for _ in range(n_steps):
        grad = score_of(x)
        noise = torch.randn_like(x)
        sums = torch.stack([ops.row_norms(grad).sum(), ops.row_norms(noise).sum(),
                            torch.tensor(float(x.shape[0]), device=x.device)])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(sums, group=group)
        g_sum, n_sum, b_tot = (float(v) for v in sums.tolist())
        step = (snr * (n_sum / b_tot) / (g_sum / b_tot)) ** 2 * 2 * alpha
        x, x_mean = ops.affine_noise_step(x, grad, noise, 1.0, step, (2 * step) ** 0.5)
    return x, x_mean


@register_corrector(name='langevin_global')
class LangevinCorrectorGlobal(Corrector):
    """``langevin`` with global-batch norms across ranks (not in the reference: replaces running it in ONE process)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        return _langevin_global(self.sde, lambda v: self.score_fn(v, t), x, self.snr, self.n_steps, t)


@register_corrector(name='conditional_langevin_global')
class conditionalLangevinCorrectorGlobal(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.cVESDE, sde_lib.cVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, y, t):
        return _langevin_global(self.sde, lambda v: self.score_fn(v, y, t), x, self.snr, self.n_steps, t)


def _ald(sde, score_of, x, t, snr, n_steps):
    """Annealed Langevin dynamics (sampling/correctors.py:111-142): step = (snr*std(t))^2 * 2*alpha,
    x_mean = x + step*grad, x = x_mean + sqrt(2*step)*z; no batch coupling."""
    t1 = t.detach().cpu().flatten()[:1].to(torch.float32)
    if t.numel() > 1 and not bool(torch.all(t == t.flatten()[0])):
        raise NotImplementedError('per-sample t within one update is not supported by the HIP step kernel')
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.cVPSDE, sde_lib.subVPSDE)):
        alpha = float(sde.alphas.to(torch.float32)[int((t1 * (sde.N - 1) / sde.T).long()[0])])
    else:
        alpha = 1.0
    std = float(sde.marginal_prob(torch.zeros(1, 1, 1, 1), t1)[1].flatten()[0])
    step = (snr * std) ** 2 * 2 * alpha
    x = x.clone()
    x_mean = x
    for _ in range(n_steps):
        grad = score_of(x)
        noise = torch.randn_like(x)
        x, x_mean = ops.affine_noise_step(x, grad, noise, 1.0, step, (2 * step) ** 0.5)
    return x, x_mean


@register_corrector(name='ald')
class AnnealedLangevinDynamics(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        return _ald(self.sde, lambda v: self.score_fn(v, t), x, t, self.snr, self.n_steps)


@register_corrector(name='conditional_ald')
class conditionalAnnealedLangevinDynamics(Corrector):
    """Not registered by the reference (only the unconditional ``ald`` is); provided for the conditional loop."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.cVESDE, sde_lib.cVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, y, t):
        return _ald(self.sde, lambda v: self.score_fn(v, y, t), x, t, self.snr, self.n_steps)
