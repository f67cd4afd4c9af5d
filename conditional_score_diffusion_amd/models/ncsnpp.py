"""NCSN++ score networks (reference models/ncsnpp.py:39-401) on the per-operator C ABI.

Registers ``ncsnpp`` and ``ncsnpp_paired`` with the reference's constructor / call signatures and the reference's
``state_dict`` keys (``all_modules.{i}.…``).  The module list is rebuilt from the config exactly as
``NCSNpp.__init__`` does (:44-236) and ``forward`` follows ``NCSNpp.forward`` (:238-388) step for step; every
tensor operation is a HIP kernel behind ``include/csd.h`` (conv, GroupNorm(+act), attention, upfirdn2d FIR
resampling, Linear, Fourier / positional embedding, axpby, bias add) - torch only allocates buffers, concatenates
skip connections (a copy) and slices views.  There is no PyTorch fallback.

This first NCSN++ executor is *operator-granular* (one C-ABI call per layer, NCHW at the boundary, weights packed
per call): it establishes parity for the family the north star names; the planned NHWC graph executor that the
DDPM family already has (csrc/unet.hip) is the next step for it.

Covered options - every NCSN++ / DDPM++ config of the reference (configs/{ve,vp,subvp}/*ncsnpp*, *ddpmpp*) falls in this set:
``resblock_type='biggan'``, ``fir`` True (any FIR kernel) or False (naive nearest / 2x2-mean resampling), ``progressive`` in
{none, output_skip}, ``progressive_input`` in {none, input_skip, residual}, ``progressive_combine='sum'``, ``embedding_type``
in {positional, fourier}, ``skip_rescale`` either
way, attention at any resolutions.  Other values raise NotImplementedError (never a silent fallback).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import require_gpu_tensor
from . import utils
from .ddpm import HipUNet, _Node, _fan_avg_uniform


def _groups(c):
    return min(c // 4, 32)          # models/layerspp.py:67,219,231; ncsnpp.py:200-233


class NCSNpp(nn.Module):
    """``ncsnpp`` (models/ncsnpp.py:39-236)."""

    def __init__(self, config, precision=None):
        super().__init__()
        m, d = config.model, config.data
        get = (lambda k, dflt=None: m.get(k, dflt)) if hasattr(m, 'get') else (lambda k, dflt=None: getattr(m, k, dflt))
        if precision is None:
            precision = get('csd_precision') or os.environ.get('CSD_PRECISION', 'fp16x3')
        if precision not in _lib.PREC_IDS:
            raise ValueError('unknown csd precision %r' % (precision,))
        self.precision = precision
        self.config = config
        self.act = m.nonlinearity.lower()
        if self.act not in _lib.ACT_IDS or self.act == 'none':
            raise NotImplementedError('activation function does not exist!')
        self.nf = nf = m.nf
        ch_mult = tuple(m.ch_mult)
        self.num_res_blocks = m.num_res_blocks
        self.attn_resolutions = tuple(m.attn_resolutions)
        self.num_resolutions = len(ch_mult)
        self.all_resolutions = [d.effective_image_size // (2 ** i) for i in range(self.num_resolutions)]
        self.conditional = bool(m.conditional)
        self.centered = bool(d.centered)
        self.fir = bool(m.fir)
        self.fir_kernel = tuple(m.fir_kernel)
        self.skip_rescale = bool(m.skip_rescale)
        self.resblock_type = m.resblock_type.lower()
        self.progressive = m.progressive.lower()
        self.progressive_input = m.progressive_input.lower()
        self.embedding_type = m.embedding_type.lower()
        self.init_scale = m.init_scale
        combine = m.progressive_combine.lower()
        assert self.progressive in ['none', 'output_skip', 'residual']
        assert self.progressive_input in ['none', 'input_skip', 'residual']
        assert self.embedding_type in ['fourier', 'positional']
        if self.resblock_type != 'biggan':
            raise NotImplementedError("ncsnpp on the HIP path: resblock_type 'biggan' only (got %r)" % self.resblock_type)
        if self.progressive == 'residual':
            # (its pyramid_upsample is up_or_down_sampling.Conv2d(up=True) = upsample_conv_2d, which fails upstream; no config uses it)
            raise NotImplementedError("ncsnpp on the HIP path: progressive='residual' is not provided")
        if combine != 'sum':
            raise NotImplementedError("ncsnpp on the HIP path: progressive_combine 'sum' only (got %r)" % combine)
        channels = d.num_channels
        self.channels = channels

        # ---- module list, in the order of NCSNpp.__init__ ----
        mods = []          # (kind, dict)
        if self.embedding_type == 'fourier':
            assert config.training.continuous, "Fourier features are only used for continuous training."
            mods.append(('fourier', dict(size=nf, scale=m.fourier_scale)))
            embed_dim = 2 * nf
        else:
            embed_dim = nf
        if self.conditional:
            mods.append(('linear', dict(cin=embed_dim, cout=nf * 4)))
            mods.append(('linear', dict(cin=nf * 4, cout=nf * 4)))
        mods.append(('conv3', dict(cin=channels, cout=nf, init_scale=1.)))
        hs_c = [nf]
        in_ch = nf
        input_pyramid_ch = channels
        for i_level in range(self.num_resolutions):
            for _ in range(self.num_res_blocks):
                out_ch = nf * ch_mult[i_level]
                mods.append(('res', dict(cin=in_ch, cout=out_ch, up=False, down=False)))
                in_ch = out_ch
                if self.all_resolutions[i_level] in self.attn_resolutions:
                    mods.append(('attn', dict(c=in_ch)))
                hs_c.append(in_ch)
            if i_level != self.num_resolutions - 1:
                mods.append(('res', dict(cin=in_ch, cout=in_ch, up=False, down=True)))
                if self.progressive_input == 'input_skip':
                    mods.append(('combine', dict(dim1=input_pyramid_ch, dim2=in_ch)))
                elif self.progressive_input == 'residual':
                    mods.append(('pyr_down', dict(cin=input_pyramid_ch, cout=in_ch)))     # ncsnpp.py:171-173
                    input_pyramid_ch = in_ch
                hs_c.append(in_ch)
        in_ch = hs_c[-1]
        mods.append(('res', dict(cin=in_ch, cout=in_ch, up=False, down=False)))
        mods.append(('attn', dict(c=in_ch)))
        mods.append(('res', dict(cin=in_ch, cout=in_ch, up=False, down=False)))
        for i_level in reversed(range(self.num_resolutions)):
            for _ in range(self.num_res_blocks + 1):
                out_ch = nf * ch_mult[i_level]
                mods.append(('res', dict(cin=in_ch + hs_c.pop(), cout=out_ch, up=False, down=False)))
                in_ch = out_ch
            if self.all_resolutions[i_level] in self.attn_resolutions:
                mods.append(('attn', dict(c=in_ch)))
            if self.progressive == 'output_skip':
                mods.append(('gn', dict(c=in_ch)))
                mods.append(('conv3', dict(cin=in_ch, cout=channels, init_scale=self.init_scale)))
            if i_level != 0:
                mods.append(('res', dict(cin=in_ch, cout=in_ch, up=True, down=False)))
        assert not hs_c
        if self.progressive != 'output_skip':
            mods.append(('gn', dict(c=in_ch)))
            mods.append(('conv3', dict(cin=in_ch, cout=channels, init_scale=self.init_scale)))
        self._mods = mods
        self.all_modules = nn.ModuleList([self._make_node(k, a) for k, a in mods])
        self._fir_cache = {}
        self._dropout_p = float(get('dropout', 0.0) or 0.0)
        self.dropout_seed = int(getattr(config, 'seed', 0) or 0)
        self._train_calls = 0
        self._drop_count = 0

    # ---- parameters: names, shapes and initialisation of the reference ----
    def _make_node(self, kind, a):
        node = _Node()

        def P(name, val, grad=True):
            node.register_parameter(name, nn.Parameter(val, requires_grad=grad))

        def conv(sub, cout, cin, k, scale):
            c = _Node()
            c.register_parameter('weight', nn.Parameter(_fan_avg_uniform((cout, cin, k, k), scale, False)))
            c.register_parameter('bias', nn.Parameter(torch.zeros(cout)))
            node.add_module(sub, c)

        def gn(sub, c):
            g = _Node()
            g.register_parameter('weight', nn.Parameter(torch.ones(c)))
            g.register_parameter('bias', nn.Parameter(torch.zeros(c)))
            node.add_module(sub, g)

        if kind == 'fourier':
            P('W', torch.randn(a['size']) * a['scale'], grad=False)          # layerspp.py:37
        elif kind == 'linear':
            P('weight', _fan_avg_uniform((a['cout'], a['cin']), 1., False))  # ncsnpp.py:97-101
            P('bias', torch.zeros(a['cout']))
        elif kind == 'conv3':
            P('weight', _fan_avg_uniform((a['cout'], a['cin'], 3, 3), a['init_scale'], False))
            P('bias', torch.zeros(a['cout']))
        elif kind == 'gn':
            P('weight', torch.ones(a['c']))
            P('bias', torch.zeros(a['c']))
        elif kind == 'combine':
            conv('Conv_0', a['dim2'], a['dim1'], 1, 1.)                      # layerspp.py:49
        elif kind == 'pyr_down':                                             # layerspp.Downsample(with_conv=True), :130-147
            conv('Conv2d_0' if self.fir else 'Conv_0', a['cout'], a['cin'], 3, 1.)
        elif kind == 'attn':
            gn('GroupNorm_0', a['c'])
            for j in range(4):
                nin = _Node()
                scale = self.init_scale if j == 3 else 0.1                   # layerspp.py:69-72; NIN default 0.1
                nin.register_parameter('W', nn.Parameter(_fan_avg_uniform((a['c'], a['c']), scale, True)))
                nin.register_parameter('b', nn.Parameter(torch.zeros(a['c'])))
                node.add_module('NIN_%d' % j, nin)
        elif kind == 'res':
            cin, cout = a['cin'], a['cout']
            gn('GroupNorm_0', cin)
            conv('Conv_0', cout, cin, 3, 1.)
            if self.conditional:
                dn = _Node()
                dn.register_parameter('weight', nn.Parameter(_fan_avg_uniform((cout, self.nf * 4), 1., False)))
                dn.register_parameter('bias', nn.Parameter(torch.zeros(cout)))
                node.add_module('Dense_0', dn)
            gn('GroupNorm_1', cout)
            conv('Conv_1', cout, cout, 3, self.init_scale)
            if cin != cout or a['up'] or a['down']:
                conv('Conv_2', cout, cin, 1, 1.)
        else:
            raise AssertionError(kind)
        return node

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- FIR resampling: up_or_down_sampling.py:181-257 (_setup_kernel, upsample_2d, downsample_2d) ----
    def _fir(self, device, gain):
        key = (str(device), gain)
        if key not in self._fir_cache:
            k = np.asarray(self.fir_kernel, dtype=np.float32)
            k = np.outer(k, k)
            k /= np.sum(k)
            self._fir_cache[key] = torch.tensor(k * gain, device=device)
        return self._fir_cache[key]

    def _box(self, device, gain):
        key = ('box', str(device), gain)
        if key not in self._fir_cache:
            self._fir_cache[key] = torch.full((2, 2), gain, dtype=torch.float32, device=device)
        return self._fir_cache[key]

    def _upsample_2d(self, x, factor=2):
        if not self.fir:       # naive_upsample_2d (up_or_down_sampling.py:59-63) = zero insertion + 2x2 box: nearest neighbour
            return ops.upfirdn2d(x, self._box(x.device, 1.0), up=2, pad=(1, 0))
        k = self._fir(x.device, float(factor ** 2))
        p = k.shape[0] - factor
        return ops.upfirdn2d(x, k, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))

    def _downsample_2d(self, x, factor=2):
        if not self.fir:       # naive_downsample_2d (:66-69): mean of 2x2 blocks
            return ops.upfirdn2d(x, self._box(x.device, 0.25), down=2, pad=(0, 0))
        k = self._fir(x.device, 1.0)
        p = k.shape[0] - factor
        return ops.upfirdn2d(x, k, down=factor, pad=((p + 1) // 2, p // 2))

    def _pyr_down(self, node, x):
        """layerspp.Downsample(with_conv=True) of the 'residual' input pyramid (layerspp.py:130-165).  fir: conv_downsample_2d =
        FIR, then a VALID stride-2 3x3 conv on the (H+1)-wide result; run as the library's stride-2 convolution (pad (0,1,0,1))
        on an (H+2)-wide FIR output - its first H/2 rows/columns read exactly the valid window - and cropped."""
        O = self._O
        if not self.fir:
            return O.conv2d(x, node.Conv_0.weight, node.Conv_0.bias, stride=2, downsample_pad=True, precision=self.precision)
        k = self._fir(x.device, 1.0)
        p = (k.shape[0] - 2) + 2
        z = ops.upfirdn2d(x, k, pad=((p + 1) // 2, p // 2 + 1))
        y = O.conv2d(z, node.Conv2d_0.weight, node.Conv2d_0.bias, stride=2, downsample_pad=True, precision=self.precision)
        return y[:, :, :-1, :-1].contiguous()

    # ---- blocks ----
    @property
    def _O(self):
        """operator set: the differentiable shells of grad_ops (HIP forward + backward kernels) while training under autograd,
        the plain forward wrappers otherwise"""
        if self.training and torch.is_grad_enabled():
            from .. import grad_ops
            return grad_ops
        return ops

    def _dropout(self, h):
        """Dropout_0 (models/layerspp.py:233,265): active only in training mode"""
        if not (self.training and torch.is_grad_enabled()) or self._dropout_p <= 0:
            return h
        from .. import grad_ops
        self._drop_count += 1
        return grad_ops.dropout(h, self._dropout_p, self.dropout_seed, (self._train_calls << 16) + self._drop_count)

    def _conv(self, node, x, ksize):
        return self._O.conv2d(x, node.weight, node.bias, precision=self.precision if ksize == 3 else 'fp32')

    def _res(self, node, a, x, temb):
        """ResnetBlockBigGANpp.forward (models/layerspp.py:242-274)."""
        cin, cout = a['cin'], a['cout']
        O = self._O
        h = O.groupnorm_act(x, node.GroupNorm_0.weight, node.GroupNorm_0.bias, groups=_groups(cin), act=self.act)
        if a['up']:
            h, x = self._upsample_2d(h), self._upsample_2d(x)
        elif a['down']:
            h, x = self._downsample_2d(h), self._downsample_2d(x)
        h = self._conv(node.Conv_0, h, 3)
        if temb is not None:
            h = O.bias_add_nchw(h, O.linear(temb, node.Dense_0.weight, node.Dense_0.bias, act_in=self.act))
        h = O.groupnorm_act(h, node.GroupNorm_1.weight, node.GroupNorm_1.bias, groups=_groups(cout), act=self.act)
        h = self._conv(node.Conv_1, self._dropout(h), 3)
        if cin != cout or a['up'] or a['down']:
            x = self._conv(node.Conv_2, x, 1)
        return O.axpby(x, h, post=(1.0 / math.sqrt(2.0)) if self.skip_rescale else 1.0)

    def _attn(self, node, a, x):
        """AttnBlockpp.forward (models/layerspp.py:75-91)."""
        C = a['c']
        O = self._O
        h = O.groupnorm_act(x, node.GroupNorm_0.weight, node.GroupNorm_0.bias, groups=_groups(C), act='none')

        def nin(n, t):        # NIN = per-pixel matmul with W [in, out] (models/layers.py:555-564) = 1x1 conv with W^T
            return O.conv2d(t, n.W.t().reshape(C, C, 1, 1).contiguous(), n.b, precision='fp32')

        q, k, v = nin(node.NIN_0, h), nin(node.NIN_1, h), nin(node.NIN_2, h)
        h = nin(node.NIN_3, O.attention(q, k, v))
        return O.axpby(x, h, post=(1.0 / math.sqrt(2.0)) if self.skip_rescale else 1.0)

    # ---- forward: NCSNpp.forward (models/ncsnpp.py:238-388) ----
    def forward(self, x, time_cond):
        require_gpu_tensor(x, 'x')
        require_gpu_tensor(time_cond, 'time_cond')
        x = x.contiguous().float()
        time_cond = time_cond.contiguous().float()
        mods, nodes = self._mods, self.all_modules
        O = self._O
        self._train_calls += 1
        self._drop_count = 0
        i = 0
        if self.embedding_type == 'fourier':
            temb = ops.fourier_embedding(time_cond, nodes[i].W)
            i += 1
        else:
            temb = ops.timestep_embedding(time_cond, self.nf)
        if self.conditional:
            temb = O.linear(temb, nodes[i].weight, nodes[i].bias)
            i += 1
            temb = O.linear(temb, nodes[i].weight, nodes[i].bias, act_in=self.act)
            i += 1
        else:
            temb = None
        if not self.centered:
            x = O.axpby(x, None, alpha=2.0, gamma=-1.0)
        input_pyramid = x if self.progressive_input != 'none' else None
        hs = [self._conv(nodes[i], x, 3)]
        i += 1
        for i_level in range(self.num_resolutions):
            for _ in range(self.num_res_blocks):
                h = self._res(nodes[i], mods[i][1], hs[-1], temb)
                i += 1
                if h.shape[-1] in self.attn_resolutions:
                    h = self._attn(nodes[i], mods[i][1], h)
                    i += 1
                hs.append(h)
            if i_level != self.num_resolutions - 1:
                h = self._res(nodes[i], mods[i][1], hs[-1], temb)
                i += 1
                if self.progressive_input == 'input_skip':
                    input_pyramid = self._downsample_2d(input_pyramid)
                    # Combine 'sum' (models/layerspp.py:53-57): Conv_0(input_pyramid) + h
                    h = O.axpby(self._conv(nodes[i].Conv_0, input_pyramid, 1), h)
                    i += 1
                elif self.progressive_input == 'residual':           # ncsnpp.py:302-309
                    input_pyramid = self._pyr_down(nodes[i], input_pyramid)
                    i += 1
                    h = O.axpby(input_pyramid, h, post=(1.0 / math.sqrt(2.0)) if self.skip_rescale else 1.0)
                    input_pyramid = h
                hs.append(h)
        h = hs[-1]
        h = self._res(nodes[i], mods[i][1], h, temb)
        i += 1
        h = self._attn(nodes[i], mods[i][1], h)
        i += 1
        h = self._res(nodes[i], mods[i][1], h, temb)
        i += 1
        pyramid = None
        for i_level in reversed(range(self.num_resolutions)):
            for _ in range(self.num_res_blocks + 1):
                h = self._res(nodes[i], mods[i][1], torch.cat([h, hs.pop()], dim=1), temb)
                i += 1
            if h.shape[-1] in self.attn_resolutions:
                h = self._attn(nodes[i], mods[i][1], h)
                i += 1
            if self.progressive == 'output_skip':
                ph = O.groupnorm_act(h, nodes[i].weight, nodes[i].bias, groups=_groups(mods[i][1]['c']), act=self.act)
                i += 1
                ph = self._conv(nodes[i], ph, 3)
                i += 1
                pyramid = ph if pyramid is None else O.axpby(self._upsample_2d(pyramid), ph)
            if i_level != 0:
                h = self._res(nodes[i], mods[i][1], h, temb)
                i += 1
        assert not hs
        if self.progressive == 'output_skip':
            h = pyramid
        else:
            h = O.groupnorm_act(h, nodes[i].weight, nodes[i].bias, groups=_groups(mods[i][1]['c']), act=self.act)
            i += 1
            h = self._conv(nodes[i], h, 3)
            i += 1
        assert i == len(nodes)
        return h




class NCSNpp_paired(NCSNpp):
    """``ncsnpp_paired`` (models/ncsnpp.py:390-401): concatenates x and y, returns both halves."""

    def forward(self, input_dict, labels):
        x, y = input_dict['x'], input_dict['y']
        xc = x.size(1)
        out = super().forward(torch.cat((x, y), dim=1), labels)
        return {'x': out[:, :xc], 'y': out[:, xc:]}




# ------------------------------------------------------------------------------------------------------------------
# Planned-graph executor (csrc/unet.hip, arch 1): the same NHWC kernels, fused GroupNorm statistics, fp16-MFMA conv
# schedules and fused PC loop as the DDPM family.  These are the classes the registry hands out; the operator-granular
# classes above stay available as ``ncsnpp_ops`` / ``ncsnpp_paired_ops`` (A/B reference for the graph executor).
# ------------------------------------------------------------------------------------------------------------------
class HipNCSNpp(HipUNet):
    arch = 1

    def __init__(self, config, precision=None):
        m = config.model
        if m.resblock_type.lower() != 'biggan':
            raise NotImplementedError("ncsnpp on the HIP path: resblock_type 'biggan' only (got %r)" % m.resblock_type)
        if m.progressive.lower() == 'residual' or m.progressive_input.lower() == 'residual':
            raise NotImplementedError("the planned NCSN++ graph does not cover 'residual' progressive growing (the registry hands "
                                      "out the operator-granular class for progressive_input='residual')")
        if m.progressive_combine.lower() != 'sum':
            raise NotImplementedError("ncsnpp on the HIP path: progressive_combine 'sum' only")
        if not m.conditional:
            raise NotImplementedError('ncsnpp on the HIP path: time-conditional networks only')
        assert m.progressive.lower() in ['none', 'output_skip'] and m.progressive_input.lower() in ['none', 'input_skip']
        assert m.embedding_type.lower() in ['fourier', 'positional']
        if m.embedding_type.lower() == 'fourier':
            assert config.training.continuous, "Fourier features are only used for continuous training."
        self.embedding_type = m.embedding_type.lower()
        self._fourier_scale = float(m.get('fourier_scale', 16) if hasattr(m, 'get') else getattr(m, 'fourier_scale', 16))
        self._init_scale = float(m.init_scale)
        self._pyramid_out = m.progressive.lower() == 'output_skip'
        self._config = config
        super().__init__(config, precision)
        self._reinit_special()

    def _out_channels(self, config):
        return config.data.num_channels

    def _extra_config(self, cfg, config):
        m = config.model
        cfg.skip_rescale = int(bool(m.skip_rescale))
        cfg.progressive = 1 if m.progressive.lower() == 'output_skip' else 0
        cfg.progressive_input = 1 if m.progressive_input.lower() == 'input_skip' else 0
        cfg.embedding_type = 1 if m.embedding_type.lower() == 'fourier' else 0
        # fir = False (naive nearest / 2x2-mean resampling, up_or_down_sampling.py:59-69) IS the 4-tap FIR (0, 1, 1, 0): the
        # normalised kernel is a 2x2 box, with the same pads
        taps = list(m.fir_kernel) if m.fir else [0., 1., 1., 0.]
        if len(taps) != 4:
            raise NotImplementedError('ncsnpp on the HIP path: 4-tap FIR kernels only (got %r)' % (taps,))
        cfg.n_fir = 4
        for i, v in enumerate(taps):
            cfg.fir_kernel[i] = float(v)

    def _train_forward(self, x, y, labels):
        """training mode under autograd.  ``train_executor == 'planned'`` (default): ONE autograd node over csd_unet_train_forward /
        csd_unet_backward (csrc/train_graph.h, arch 1: BigGAN blocks with FIR resampling, Combine, pyramids - every parameter
        gradient from one call).  ``'operators'``: the operator-granular NCSN++ (class NCSNpp above, differentiable HIP operators) on
        THIS model's parameters - a twin whose nn.Parameters are the same objects, so gradients land in ``self.parameters()``."""
        if self.train_executor == 'planned':
            return self._train_forward_planned(x, y, labels)
        twin = self.__dict__.get('_twin')
        if twin is None:
            twin = NCSNpp(self._config, precision=self.precision)
            mine = dict(self.named_parameters())
            for name, _ in list(twin.named_parameters()):
                mod = twin
                parts = name.split('.')
                for part in parts[:-1]:
                    mod = getattr(mod, part)
                setattr(mod, parts[-1], mine[name])
            assert {k for k, _ in twin.named_parameters()} == set(mine)
            self.__dict__['_twin'] = twin            # (not a registered submodule: the state_dict stays the reference's)
        twin.train()
        twin.dropout_seed = self.dropout_seed
        # THIS model owns the dropout stream position (Trainer.state_dict saves it): the twin's forward increments its own copy
        twin._train_calls = self._train_calls
        self._train_calls += 1
        inp = torch.cat([x, y], dim=1) if self.y_channels else x
        return twin(inp, labels)

    def _reinit_special(self):
        """initialisation the generic rules of HipUNet._build_params do not cover: the Gaussian Fourier W
        (layerspp.py:37) and the init_scale of the pyramid / output convolutions (ncsnpp.py:203-233)."""
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                if k.endswith('.W') and v.dim() == 1:
                    v.copy_(torch.randn(v.shape) * self._fourier_scale)
                    v.requires_grad_(False)
                parts = k.split('.')
                if len(parts) == 3 and parts[2] == 'weight' and v.dim() == 4 and v.shape[0] == self.out_channels:
                    v.copy_(_fan_avg_uniform(tuple(v.shape), self._init_scale, False))


class NCSNppPlanned(HipNCSNpp):
    """``ncsnpp`` (models/ncsnpp.py:39-388): ``model(x, time_cond) -> Tensor``."""

    def _channels(self, config):
        return int(config.data.num_channels), 0

    def forward(self, x, time_cond):
        return self._run(x, None, time_cond)


class NCSNppPairedPlanned(HipNCSNpp):
    """``ncsnpp_paired`` (models/ncsnpp.py:390-401): ``model({'x','y'}, labels) -> {'x','y'}``."""

    def _channels(self, config):
        d = config.data
        cx = int(d.shape_x[0]) if ('shape_x' in d if hasattr(d, '__contains__') else hasattr(d, 'shape_x')) else int(d.num_channels) // 2
        return cx, int(d.num_channels) - cx

    def forward(self, input_dict, labels):
        out = self._run(input_dict['x'], input_dict['y'], labels)
        c = self.x_channels
        return {'x': out[:, :c], 'y': out[:, c:]}


def _needs_operator_granular(config):
    return config.model.progressive_input.lower() == 'residual'


def create_ncsnpp(config, **kw):
    """``ncsnpp``: the planned graph executor, or the operator-granular class for the options only it covers"""
    return (NCSNpp if _needs_operator_granular(config) else NCSNppPlanned)(config, **kw)


def create_ncsnpp_paired(config, **kw):
    return (NCSNpp_paired if _needs_operator_granular(config) else NCSNppPairedPlanned)(config, **kw)


utils.register_model(create_ncsnpp, name='ncsnpp')
utils.register_model(create_ncsnpp_paired, name='ncsnpp_paired')
utils.register_model(NCSNpp, name='ncsnpp_ops')
utils.register_model(NCSNpp_paired, name='ncsnpp_paired_ops')
