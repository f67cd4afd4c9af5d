"""ORACLE tooling (build container only): import the *reference* repo from /root/reference.

The reference needs pytorch_lightning, torchvision, ml_collections and a CUDA JIT build at import
time.  None exist here, so four stand-ins are installed in
``sys.modules`` BEFORE the reference modules are imported.  They replace *missing third-party
packages* only - every reference source file is imported unmodified from where it lies.
Nothing in here runs on the GPU box (/root/reference does not exist there).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get('CSD_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'models'))


def install():
    if not available():
        raise RuntimeError('reference checkout not found at %s' % REF)
    if 'pytorch_lightning' not in sys.modules:
        pl = types.ModuleType('pytorch_lightning')

        class LightningModule(nn.Module):
            @property
            def device(self):
                return next(self.parameters()).device

        pl.LightningModule = LightningModule
        sys.modules['pytorch_lightning'] = pl
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tr = types.ModuleType('torchvision.transforms')
        fn = types.ModuleType('torchvision.transforms.functional')

        class InterpolationMode:
            NEAREST = 'nearest'
            BICUBIC = 'bicubic'

        class Resize:
            def __init__(self, *a, **k):
                pass

        fn.InterpolationMode = InterpolationMode
        tr.Resize = Resize
        tr.functional = fn
        tr.InterpolationMode = InterpolationMode
        tv.transforms = tr
        sys.modules.update({'torchvision': tv, 'torchvision.transforms': tr,
                            'torchvision.transforms.functional': fn})
    if 'ml_collections' not in sys.modules:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from conditional_score_diffusion_amd.config_dict import ConfigDict
        ml = types.ModuleType('ml_collections')
        ml.ConfigDict = ConfigDict
        sys.modules['ml_collections'] = ml
    # op/upfirdn2d.py:12 and op/fused_act.py:11 JIT-compile CUDA at import; stub the loader so the
    # CPU branches (upfirdn2d_native, F.leaky_relu) are the ones exercised.
    import torch.utils.cpp_extension as cpp
    if not getattr(cpp, '_csd_stubbed', False):
        cpp.load = lambda *a, **k: None
        cpp._csd_stubbed = True
    if REF not in sys.path:
        sys.path.insert(0, REF)


def modules():
    """Import and return the reference modules on the hot path."""
    install()
    import importlib
    names = ['sde_lib', 'models.utils', 'models.layers', 'models.ddpm', 'models.layerspp',
             'models.up_or_down_sampling', 'models.ncsnpp', 'sampling.predictors',
             'sampling.correctors', 'sampling.conditional', 'sampling.unconditional', 'losses']
    out = {}
    for n in names:
        out[n] = importlib.import_module(n)
    return out


class TapeRandn:
    """Context manager: torch.randn / torch.randn_like read from a list."""

    def __init__(self, tensors):
        self.t, self.i = list(tensors), 0

    def _next(self, shape):
        z = self.t[self.i]
        self.i += 1
        assert tuple(z.shape) == tuple(shape), (z.shape, shape)
        return z.clone()

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        torch.randn = lambda *s, **k: self._next(s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
        torch.randn_like = lambda x, **k: self._next(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._randn, self._randn_like
