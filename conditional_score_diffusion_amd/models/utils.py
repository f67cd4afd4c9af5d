"""Model registry and score-function glue (host side).

Mirrors the public surface of the reference ``models/utils.py``: ``register_model`` /
``get_model`` / ``create_model`` (models/utils.py:24-47,114-120), ``get_model_fn`` (:123-152),
``divide_by_sigmas`` (:50-74), ``get_score_fn`` (:156-267), ``get_conditional_score_fn``
(:270-278) and the small helpers (:76-111, :280-286).  The registry here is this package's own
(never shared with an imported reference: duplicate names raise ValueError there, :35-36).

The only tensor arithmetic at this level - dividing the network output by sigma(t) - goes through
the HIP kernel ``csd_scale_rows`` for GPU tensors.
"""
import numpy as np
import torch

from .. import ops, sde_lib

_MODELS = {}


def register_model(cls=None, *, name=None):
    """Decorator registering a model class under ``name`` (default: the class name)."""

    def _register(c):
        key = c.__name__ if name is None else name
        if key in _MODELS:
            raise ValueError(f'Already registered model with name: {key}')
        _MODELS[key] = c
        return c

    return _register if cls is None else _register(cls)


def get_model(name):
    return _MODELS[name]


def create_model(config):
    """``config.model.name`` -> instance (models/utils.py:114-120)."""
    return get_model(config.model.name)(config)


def get_sigmas(config):
    """SMLD noise levels, descending (models/utils.py:76-88)."""
    return np.exp(np.linspace(np.log(config.model.sigma_max), np.log(config.model.sigma_min),
                              config.model.num_scales))


def get_ddpm_params(config):
    """DDPM beta/alpha tables (models/utils.py:91-111)."""
    n = 1000
    b0 = config.model.beta_min / config.model.num_scales
    b1 = config.model.beta_max / config.model.num_scales
    betas = np.linspace(b0, b1, n, dtype=np.float64)
    alphas = 1. - betas
    ac = np.cumprod(alphas, axis=0)
    return {'betas': betas, 'alphas': alphas, 'alphas_cumprod': ac, 'sqrt_alphas_cumprod': np.sqrt(ac),
            'sqrt_1m_alphas_cumprod': np.sqrt(1. - ac), 'beta_min': b0 * (n - 1), 'beta_max': b1 * (n - 1),
            'num_diffusion_timesteps': n}


def get_model_fn(model, train=False):
    """(x, labels) -> model output, switching eval/train mode on every call like the reference."""

    def model_fn(x, labels):
        model.train() if train else model.eval()
        return model(x, labels)

    return model_fn


def _div_rows(h, denom):
    """h / denom[b] with the HIP row-scale kernel on GPU tensors (host torch only for CPU inputs,
    e.g. the loss-side bookkeeping of tiny tensors)."""
    if h.is_cuda and h.requires_grad:
        from .. import grad_ops
        return grad_ops.scale_rows(h, denom.to(device=h.device, dtype=torch.float32).contiguous(), divide=True)
    if h.is_cuda:
        return ops.scale_rows(h, denom.to(device=h.device, dtype=torch.float32).contiguous(), divide=True)
    return h / denom[(...,) + (None,) * (h.dim() - 1)]


def divide_by_sigmas(h, labels, sde, continuous=False):
    """Scale raw network output(s) by 1/sigma (models/utils.py:50-74). ``h`` and ``sde`` may be dicts."""
    def sig(s, like):
        if continuous:
            return s.marginal_prob(torch.zeros((labels.shape[0],) + (1,) * (like.dim() - 1),
                                               device=labels.device), labels)[1]
        return s.discrete_sigmas.to(labels.device).type_as(like)[labels]

    if isinstance(sde, dict) and isinstance(h, dict):
        return {k: _div_rows(v, sig(sde[k], v)) for k, v in h.items()}
    return _div_rows(h, sig(sde, h))


def get_score_fn(sde, model, conditional=False, train=False, continuous=False):
    """Wrap a network into a time-dependent score function (models/utils.py:156-267).

    Branches are selected by SDE class exactly as the reference does; unsupported combinations
    raise NotImplementedError with the same wording."""
    model_fn = get_model_fn(model, train=train)
    ve = (sde_lib.VESDE, sde_lib.cVESDE)

    if conditional:
        if isinstance(sde, dict):
            if isinstance(sde['y'], (sde_lib.VPSDE, sde_lib.subVPSDE)):
                raise NotImplementedError('This combination of sdes is not supported for conditional SDEs yet.')
            if not (isinstance(sde['y'], sde_lib.VESDE) and isinstance(sde['x'], sde_lib.cVESDE) and len(sde) == 2):
                raise NotImplementedError('This combination of SDEs is not supported for conditional SDEs yet.')
            N = sde['x'].N
        elif isinstance(sde, sde_lib.cVPSDE):
            def score_fn(x, t):
                labels = t * (sde.N - 1)
                out = model_fn(x, labels)
                if continuous:
                    std = sde.marginal_prob(torch.zeros_like(t)[:, None], t)[1]
                else:
                    std = sde.sqrt_1m_alphas_cumprod.to(labels.device).type_as(labels)[labels.long()]
                return _div_rows(out, std)
            return score_fn
        elif isinstance(sde, ve):
            N = sde.N
        else:
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

        def score_fn(x, t):
            # VE, SR3-style single SDE or the {'x','y'} pair: label = t*(N-1) (models/utils.py:173-186,210-221)
            labels = t * (N - 1)
            if continuous:
                return divide_by_sigmas(model_fn(x, labels), t, sde, True)
            labels = torch.round(labels.float()).long()
            return divide_by_sigmas(model_fn(x, labels), labels, sde, False)
        return score_fn

    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        def score_fn(x, t):
            labels = t * (sde.N - 1)
            out = model_fn(x, labels)
            if continuous or isinstance(sde, sde_lib.subVPSDE):
                std = sde.marginal_prob(torch.zeros_like(t)[:, None], t)[1]
            else:
                std = sde.sqrt_1m_alphas_cumprod.to(labels.device).type_as(labels)[labels.long()]
            return _div_rows(out, std)
        return score_fn
    if isinstance(sde, ve):
        def score_fn(x, t):
            if continuous:
                # label is sigma(t) (log sigma for Fourier embeddings) (models/utils.py:246-253)
                std = sde.marginal_prob(torch.zeros_like(t)[:, None], t)[1]
                emb = torch.log(std) if getattr(model, 'embedding_type', 'positional') == 'fourier' else std
                return _div_rows(model_fn(x, emb), std)
            labels = torch.round(t * (sde.N - 1)).long()
            std = sde.discrete_sigmas.to(t.device).type_as(x)[labels]
            return _div_rows(model_fn(x, std), std)
        return score_fn
    raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


def get_conditional_score_fn(score_fn, target_domain):
    """(x, y, t) -> score of ``target_domain`` (models/utils.py:270-278)."""

    def conditional_score_fn(x, y, t):
        score = score_fn({'x': x, 'y': y}, t)
        return score[target_domain] if isinstance(score, dict) else score

    return conditional_score_fn


def to_flattened_numpy(x):
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    return torch.from_numpy(x.reshape(shape))
