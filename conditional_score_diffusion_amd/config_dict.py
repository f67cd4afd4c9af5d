"""Minimal stand-in for ``ml_collections.ConfigDict``.

The reference's config files (e.g. configs/ve/inverse_problems/super_resolution/
celebA_SR3_160.py:6-161) only ever *construct* ``ml_collections.ConfigDict()`` objects and
assign attributes on them, and the hot path only ever *reads* attributes.  ``ml_collections``
is not installed in the target image, so this attribute-dict lets those config files load
unchanged (see ``load_reference_config``) and lets the hot path be driven by identical
``config.model.*`` / ``config.sampling.*`` / ``config.data.*`` values.
"""
import importlib.util
import sys
import types


class ConfigDict(dict):
    """dict with attribute access; nested dicts are promoted on assignment."""

    def __init__(self, initial=None, **kw):
        super().__init__()
        for k, v in dict(initial or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def get(self, k, default=None):
        return self[k] if k in self else default

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}

    def copy_and_resolve_references(self):
        return ConfigDict(self.to_dict())


def load_reference_config(path):
    """Execute a reference-style config file (``def get_config(): ...``) and return its ConfigDict.

    A shim module named ``ml_collections`` exposing this ConfigDict is installed for the duration
    of the import when the real package is missing.
    """
    shim = None
    if 'ml_collections' not in sys.modules:
        shim = types.ModuleType('ml_collections')
        shim.ConfigDict = ConfigDict
        sys.modules['ml_collections'] = shim
    try:
        spec = importlib.util.spec_from_file_location('_csd_cfg', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.get_config()
    finally:
        if shim is not None:
            sys.modules.pop('ml_collections', None)
