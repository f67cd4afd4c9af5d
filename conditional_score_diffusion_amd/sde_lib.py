"""Forward SDEs, their marginals/discretisations and the reverse-time SDE factory.

Host-side algebra only: every method works on tiny ``[B]`` vectors (or is called once per
sampling step), so it stays in torch for bit-compatibility with the reference's fp32 scalar
math.  The per-pixel arithmetic that *uses* these scalars (predictor / corrector updates) lives
in the HIP library (csrc/sampler.hip).

Mirrors the public surface of the reference ``sde_lib.py``:
  SDE (sde_lib.py:7-102), cSDE (:104-142), VPSDE (:144-195), cVPSDE (:197-248),
  subVPSDE (:251-287), VESDE (:290-362), cVESDE (:364-418).
Class identity matters to callers (``isinstance`` dispatch in models/utils.py:171-253 and
sampling/correctors.py:54-56,84-86), so the same five concrete classes exist here with the same
attributes (N, T, sigma_min, sigma_max, discrete_sigmas, beta_0, beta_1, discrete_betas, alphas,
alphas_cumprod, sqrt_alphas_cumprod, sqrt_1m_alphas_cumprod).
"""
import abc

import numpy as np
import torch


def _bcast(v, like):
    """[B] -> [B,1,1,...] so it broadcasts against ``like``."""
    return v[(...,) + (None,) * (like.dim() - 1)]


class SDE(abc.ABC):
    """Abstract forward SDE on mini-batches. (reference: sde_lib.py:7-63)"""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        """End time."""

    @abc.abstractmethod
    def sde(self, x, t):
        """(drift, diffusion) of dx = f dt + g dw."""

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        """(mean, std) of p_t(x(t) | x(0)=x)."""

    @abc.abstractmethod
    def prior_sampling(self, shape):
        """One sample of p_T."""

    @abc.abstractmethod
    def prior_logp(self, z):
        """log p_T(z)."""

    def discretize(self, x, t):
        """Euler-Maruyama default: x_{i+1} = x_i + f_i + G_i z_i (sde_lib.py:49-63)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    # --- reverse-time factory -------------------------------------------------------------
    _conditional = False  # cSDE flips this: score_fn / sde / discretize take (x, y, t)

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE/ODE object (sde_lib.py:65-102 and, for cSDE, :104-142).

        The returned object is an instance of a subclass of ``type(self)`` (callers rely on
        ``isinstance(rsde, VESDE)`` etc.), exposing N, T, probability_flow, sde(), discretize().
        """
        fwd = self
        half = 0.5 if probability_flow else 1.0
        conditional = self._conditional

        class RSDE(self.__class__):
            def __init__(rs):  # noqa: N805 - deliberately skips the forward ctor
                rs.N = fwd.N
                rs.probability_flow = probability_flow

            @property
            def T(rs):  # noqa: N805
                return fwd.T

            def _split(rs, args):  # noqa: N805
                if conditional:
                    x, y, t = args
                    return x, t, (x, y, t)
                x, t = args
                return x, t, (x, t)

            def sde(rs, *args):  # noqa: N805
                x, t, sargs = rs._split(args)
                drift, diffusion = fwd.sde(x, t)
                score = score_fn(*sargs)
                drift = drift - _bcast(diffusion, x) ** 2 * score * half
                return drift, (0. if probability_flow else diffusion)

            def discretize(rs, *args):  # noqa: N805
                x, t, sargs = rs._split(args)
                f, G = fwd.discretize(x, t)
                rev_f = f - _bcast(G, x) ** 2 * score_fn(*sargs) * half
                rev_G = torch.zeros_like(G) if probability_flow else G
                return rev_f, rev_G

        return RSDE()


class cSDE(SDE):
    """Conditional setting: the score (and hence the reverse SDE) also sees ``y``."""
    _conditional = True


# ------------------------------------------------------------------------------------------
# Variance preserving family
# ------------------------------------------------------------------------------------------
class _VPMixin:
    def _vp_init(self, beta_min, beta_max, N):
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def _beta(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def _log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def sde(self, x, t):
        beta_t = self._beta(t)
        return -0.5 * _bcast(beta_t, x) * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(_bcast(lmc, x)) * x, torch.sqrt(1. - torch.exp(2. * lmc))

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.

    def discretize(self, x, t):
        """DDPM (ancestral) discretisation (sde_lib.py:186-195)."""
        i = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[i]
        alpha = self.alphas.to(x.device)[i]
        return _bcast(torch.sqrt(alpha), x) * x - x, torch.sqrt(beta)


class VPSDE(_VPMixin, SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        SDE.__init__(self, N)
        self._vp_init(beta_min, beta_max, N)


class cVPSDE(_VPMixin, cSDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        cSDE.__init__(self, N)
        self._vp_init(beta_min, beta_max, N)


class subVPSDE(SDE):
    """sub-VP SDE (sde_lib.py:251-287). No custom discretisation: Euler-Maruyama default."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        drift = -0.5 * _bcast(beta_t, x) * x
        discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return drift, torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return _bcast(torch.exp(lmc), x) * x, 1 - torch.exp(2. * lmc)

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.


# ------------------------------------------------------------------------------------------
# Variance exploding family (the one all BASELINE configs use)
# ------------------------------------------------------------------------------------------
class _VEMixin:
    def _ve_init(self, sigma_min, sigma_max, N, data_mean):
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        # fp32 table; [0]=sigma_min ... [N-1]=sigma_max (sde_lib.py:301,376)
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))
        self.N = N
        self.diffused_mean = data_mean

    @property
    def T(self):
        return 1

    def _sigma_t(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def marginal_prob(self, x, t):
        smin = torch.tensor(self.sigma_min).type_as(t)
        smax = torch.tensor(self.sigma_max).type_as(t)
        return x, smin * (smax / smin) ** t

    def prior_sampling(self, shape):
        z = torch.randn(*shape) * self.sigma_max
        if self.diffused_mean is not None:
            z = z + self.diffused_mean.unsqueeze(0).repeat(shape[0], *([1] * (len(shape) - 1)))
        return z

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return (-n / 2. * np.log(2 * np.pi * self.sigma_max ** 2)
                - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2))

    def discretize(self, x, t):
        """SMLD / NCSN discretisation: G_i = sqrt(sigma_i^2 - sigma_{i-1}^2), sigma_{-1}:=0
        (sde_lib.py:353-362, 410-418). The index is the *truncated* fp32 product t*(N-1)."""
        i = (t * (self.N - 1) / self.T).long()
        sig = self.discrete_sigmas.to(t.device)[i]
        adj = torch.where(i == 0, torch.zeros_like(t), self.discrete_sigmas[i - 1].to(t.device))
        return torch.zeros_like(x), torch.sqrt(sig ** 2 - adj ** 2)

    def compute_backward_kernel(self, x0, x_tplustau, t, tau):
        """Parameters of p(x(t) | x(0), x(t+tau)) (sde_lib.py:323-339) - used by ``use_path``."""
        smin = torch.tensor(self.sigma_min).type_as(t)
        smax = torch.tensor(self.sigma_max).type_as(t)
        s_t = (smin * (smax / smin) ** t) ** 2
        s_tt = (smin * (smax / smin) ** (t + tau)) ** 2
        std = torch.sqrt(s_t * (s_tt - s_t) / s_tt)
        w0 = (s_tt - s_t) / s_tt
        w1 = s_t / s_tt
        return x0 * _bcast(w0, x0) + x_tplustau * _bcast(w1, x0), std


class VESDE(_VEMixin, SDE):
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, data_mean=None):
        SDE.__init__(self, N)
        self._ve_init(sigma_min, sigma_max, N, data_mean)

    def sde(self, x, t):
        # constant cast to t's dtype *before* the sqrt (sde_lib.py:313)
        c = torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))).type_as(t))
        return torch.zeros_like(x), self._sigma_t(t) * c


class cVESDE(_VEMixin, cSDE):
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, data_mean=None):
        cSDE.__init__(self, N)
        self._ve_init(sigma_min, sigma_max, N, data_mean)

    def sde(self, x, t):
        # float64 0-dim constant; product with fp32 [B] stays fp32 (sde_lib.py:386-387, SURVEY App.B)
        c = torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)),
                                    device=t.device))
        return torch.zeros_like(x), self._sigma_t(t) * c
