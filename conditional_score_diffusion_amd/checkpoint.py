"""Reader for the reference's Lightning checkpoints and the module-level API around a loaded model.

A ``.ckpt`` written by the reference's trainer is a torch pickle ``{'state_dict': {...}, 'hyper_parameters': {'config': ...}, ...}``
(``BaseSdeGenerativeModel.__init__`` calls ``save_hyperparameters()``, lightning_modules/BaseSdeGenerativeModel.py:16-21).  The network
weights live under ``score_model.`` with the ``all_modules.{i}.…`` keys this package's adapters use; the VS-CMDE module also registers
the buffers ``sigma_max_y`` / ``sigma_min_y`` that its ``configure_sde`` reads back (lightning_modules/ConditionalSdeGenerativeModel.py:
136-141, 143-175).  ``load_score_module`` does what ``create_lightning_module(config, checkpoint_path)`` (lightning_modules/utils.py:
23-27) + ``configure_sde`` do: model from the stored (or given) config, weights, SDE objects - with sde['y'] rebuilt from the buffers -
and returns a ``ScoreModule`` whose ``sample()`` has the reference module's signature.

Unpickling: a checkpoint is read with ``torch.load(weights_only=True)`` first; files that carry non-tensor objects (the config) go
through a RESTRICTED unpickler that only reconstructs tensors, containers, numbers and ``ConfigDict``s (``ml_collections`` objects are
mapped onto this package's ConfigDict - the package is not in the image) and refuses every other global, so loading a file does not
execute code from it.  ``trust=True`` falls back to the plain pickle machinery for files with other objects.
"""
import collections
import pickle

import torch

from . import sde_lib
from .config_dict import ConfigDict


class _MLConfigDict(ConfigDict):
    """target of ``ml_collections.ConfigDict`` pickles: state = {'_fields': {...}, '_locked': ..., ...}"""

    def __setstate__(self, state):
        fields = state.get('_fields', state) if isinstance(state, dict) else {}
        for k, v in fields.items():
            self[k] = v.get() if hasattr(v, 'get') and type(v).__name__ == '_FieldRef' else v


class _FieldRef:
    """``ml_collections.FieldReference``: keeps its value only"""

    def __setstate__(self, state):
        self._v = state.get('_value') if isinstance(state, dict) else state

    def get(self):
        return self._v


_SAFE_BUILTINS = {('collections', 'OrderedDict'): collections.OrderedDict, ('builtins', 'dict'): dict, ('builtins', 'list'): list,
                  ('builtins', 'tuple'): tuple, ('builtins', 'set'): set, ('builtins', 'frozenset'): frozenset,
                  ('builtins', 'int'): int, ('builtins', 'float'): float, ('builtins', 'bool'): bool, ('builtins', 'str'): str,
                  ('builtins', 'bytes'): bytes, ('builtins', 'complex'): complex, ('builtins', 'slice'): slice,
                  ('builtins', 'range'): range, ('builtins', 'NoneType'): type(None)}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_BUILTINS:
            return _SAFE_BUILTINS[(module, name)]
        if module.startswith('ml_collections') and name == 'ConfigDict' or (module, name) == (ConfigDict.__module__, 'ConfigDict'):
            return _MLConfigDict if module.startswith('ml_collections') else ConfigDict
        if module.startswith('ml_collections') and name == 'FieldReference':
            return _FieldRef
        if module in ('torch._utils', 'torch', 'torch.storage', 'torch._tensor') and name in (
                '_rebuild_tensor_v2', '_rebuild_tensor', '_rebuild_parameter', '_rebuild_from_type_v2', 'FloatStorage', 'DoubleStorage',
                'HalfStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage', 'ShortStorage', 'CharStorage', 'ByteStorage',
                'BoolStorage', 'UntypedStorage', 'TypedStorage', 'Tensor', 'Size', 'device', 'float32', 'float64', 'float16',
                'bfloat16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool'):
            import importlib
            return getattr(importlib.import_module(module), name)
        if module in ('numpy', 'numpy.core.multiarray', 'numpy._core.multiarray', 'numpy.core.numeric', 'numpy._core.numeric') and name in (
                'dtype', 'ndarray', 'scalar', '_reconstruct', '_frombuffer'):
            import importlib
            return getattr(importlib.import_module(module), name)
        raise pickle.UnpicklingError('checkpoint holds a %s.%s object: refused (pass trust=True to unpickle arbitrary objects from a file '
                                     'you trust)' % (module, name))


class _RestrictedPickle:
    """the ``pickle_module`` interface torch.load expects"""
    __name__ = 'restricted_pickle'
    Unpickler = _RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def read_checkpoint(path, trust=False):
    """-> the checkpoint dict.  Tensors-only files load with ``weights_only=True``; files with a config go through the restricted
    unpickler; ``trust=True`` allows arbitrary pickled objects."""
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except Exception:
        pass
    if trust:
        return torch.load(path, map_location='cpu', weights_only=False)
    return torch.load(path, map_location='cpu', weights_only=False, pickle_module=_RestrictedPickle)


def split_lightning_state_dict(ckpt, prefix='score_model.'):
    """-> (network state_dict with the prefix stripped, {other tensor entries})."""
    sd = ckpt.get('state_dict', ckpt)
    net, rest = {}, {}
    for k, v in sd.items():
        if k.startswith(prefix):
            net[k[len(prefix):]] = v
        else:
            rest[k] = v
    if not net:
        raise KeyError('no %r entries in the checkpoint (keys start with: %s)' % (prefix, sorted({k.split(".")[0] for k in sd})))
    return net, rest


def load_lightning_checkpoint(model, path_or_ckpt, prefix='score_model.', strict=True, trust=False):
    """Load the network weights of a reference Lightning checkpoint into ``model`` (an adapter of this package).
    Returns the remaining entries (e.g. ``sigma_max_y``), which ``configure_sde`` feeds to the SDE objects."""
    ckpt = path_or_ckpt if isinstance(path_or_ckpt, dict) else read_checkpoint(path_or_ckpt, trust)
    net, rest = split_lightning_state_dict(ckpt, prefix)
    model.load_state_dict(net, strict=strict)
    return rest


def checkpoint_config(ckpt):
    """the ``config`` the Lightning module was constructed with (``save_hyperparameters``), as a ConfigDict, or None"""
    hp = ckpt.get('hyper_parameters') if isinstance(ckpt, dict) else None
    cfg = hp.get('config') if isinstance(hp, dict) else None
    if cfg is None:
        return None
    return cfg if isinstance(cfg, ConfigDict) else ConfigDict(cfg)


def configure_sde(config, buffers=None):
    """The SDE objects a reference Lightning module builds for ``config`` - BaseSdeGenerativeModel.configure_sde (:23-40) for
    unconditional models, ConditionalSdeGenerativeModel.configure_sde (:17-43) for 'sr3' / two-SDE conditioning, and the
    decreasing-variance module's (:143-175), whose conditioning SDE comes from the checkpoint's ``sigma_max_y`` / ``sigma_min_y``
    buffers.  -> (sde, sampling_eps).  (``data.use_data_mean`` needs the dataset's mean file and is not supported here.)"""
    m, name = config.model, config.training.sde.lower()
    if config.data.get('use_data_mean', False):
        raise NotImplementedError('data.use_data_mean: the data-mean prior needs the dataset statistics file')
    # which Lightning module built the checkpoint decides (lightning_modules/utils.py:23-27 create_lightning_module dispatches on
    # config.training.lightning_module: 'base' | 'conditional' | 'conditional_decreasing_variance'); configs without that key fall
    # back to the markers of a conditional model
    lm = str(config.training.get('lightning_module', '') or '').lower()
    if lm in ('base', 'conditional', 'conditional_decreasing_variance'):
        conditional = lm != 'base'
    else:
        conditional = 'conditioning_approach' in config.training or m.name.lower().endswith(('paired', 'sr3'))
    if name == 'vpsde':
        # ConditionalSdeGenerativeModel.configure_sde (:18-21): cVPSDE, CDE ('sr3') only; the decreasing-variance module (:144-146)
        # and the base module build a plain VPSDE
        if lm == 'conditional' or (not lm and conditional):
            if config.training.get('conditioning_approach', 'sr3') != 'sr3':
                raise NotImplementedError('We support only CDE with VP sde currently.')
            return sde_lib.cVPSDE(beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales), 1e-3
        return sde_lib.VPSDE(beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales), 1e-3
    if name == 'subvpsde':
        return sde_lib.subVPSDE(beta_min=m.beta_min, beta_max=m.beta_max, N=m.num_scales), 1e-3
    if name != 'vesde':
        raise NotImplementedError('SDE %s unknown.' % config.training.sde)
    if not conditional:
        return sde_lib.VESDE(sigma_min=m.sigma_min, sigma_max=m.sigma_max, N=m.num_scales), 1e-5
    sde_x = sde_lib.cVESDE(sigma_min=m.sigma_min_x, sigma_max=m.sigma_max_x, N=m.num_scales)
    if config.training.get('conditioning_approach', 'ours_NDV') == 'sr3':
        return sde_x, 1e-5
    buffers = buffers or {}
    smax = float(buffers['sigma_max_y']) if 'sigma_max_y' in buffers else float(m.sigma_max_y)
    smin = float(buffers['sigma_min_y']) if 'sigma_min_y' in buffers else float(m.sigma_min_y)
    return {'x': sde_x, 'y': sde_lib.VESDE(sigma_min=smin, sigma_max=smax, N=m.num_scales)}, 1e-5


class ScoreModule:
    """What the reference's Lightning modules are to the sampling path: ``.score_model``, ``.sde``, ``.sampling_eps``, ``.config`` and
    ``.sample(...)`` with the signatures of ConditionalSdeGenerativeModel.sample (:77-85) / BaseSdeGenerativeModel.sample (:60-66)."""

    def __init__(self, config, score_model, sde, sampling_eps, buffers=None):
        self.config, self.score_model, self.sde, self.sampling_eps = config, score_model, sde, sampling_eps
        self.buffers = dict(buffers or {})

    def to(self, device):
        self.score_model = self.score_model.to(device)
        return self

    def sample(self, y=None, show_evolution=False, num_samples=None, predictor='default', corrector='default', p_steps='default',
               c_steps='default', snr='default', denoise='default', use_path='default', **sampler_kw):
        from .sampling.conditional import get_conditional_sampling_fn
        from .sampling.unconditional import get_sampling_fn
        if y is None:
            n = num_samples if num_samples is not None else self.config.eval.batch_size
            fn = get_sampling_fn(self.config, self.sde, [n] + list(self.config.data.shape), self.sampling_eps)
            return fn(self.score_model, show_evolution=show_evolution, **sampler_kw)
        shape = [y.size(0)] + list(self.config.data.shape_x)
        fn = get_conditional_sampling_fn(config=self.config, sde=self.sde, shape=shape, eps=self.sampling_eps, predictor=predictor,
                                         corrector=corrector, p_steps=p_steps, c_steps=c_steps, snr=snr, denoise=denoise,
                                         use_path=use_path)
        return fn(self.score_model, y, show_evolution, **sampler_kw)


def load_score_module(path, config=None, device=None, trust=False, strict=True):
    """``create_lightning_module(config, checkpoint_path)`` + ``configure_sde`` for the sampling path: a ScoreModule with the
    checkpoint's weights and SDEs.  ``config`` defaults to the one stored in the checkpoint's hyper_parameters."""
    from .models import utils as mutils
    ckpt = read_checkpoint(path, trust)
    cfg = config if config is not None else checkpoint_config(ckpt)
    if cfg is None:
        raise ValueError('the checkpoint stores no hyper_parameters.config: pass the config it was trained with')
    model = mutils.create_model(cfg)
    buffers = load_lightning_checkpoint(model, ckpt, strict=strict)
    sde, eps = configure_sde(cfg, buffers)
    mod = ScoreModule(cfg, model.eval(), sde, eps, buffers)
    return mod.to(device) if device is not None else mod
