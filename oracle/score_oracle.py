"""ORACLE (test infrastructure - NOT product code).

CPU restatement, in plain functional PyTorch fp32, of the reference's score network
(DDPM-family U-Net and NCSN++) and of its predictor-corrector sampling loop.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
product path (conditional_score_diffusion_amd/) never does.

Pinning: the reference holds no golden vectors for this path, so this
oracle is pinned by fixtures generated from the *imported reference itself* in the build
container (oracle/make_goldens.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Every function cites the reference lines it follows (paths relative to /root/reference).
Parameters are passed as a flat dict with the reference's ``state_dict`` keys
(``all_modules.{i}.Conv_0.weight`` ...), conv weights OIHW, NIN ``W`` [in,out], Linear [out,in].
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------
# configuration view
# ------------------------------------------------------------------------------------------
class NetCfg:
    """The handful of values DDPM.__init__ reads (models/ddpm.py:82-147)."""

    def __init__(self, nf, ch_mult, num_res_blocks, attn_resolutions, image_size,
                 input_channels, output_channels, resamp_with_conv=True, conditional=True,
                 centered=False, act='swish'):
        self.nf = nf
        self.ch_mult = tuple(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.image_size = image_size
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.resamp_with_conv = resamp_with_conv
        self.conditional = conditional
        self.centered = centered
        self.act = act

    @staticmethod
    def from_config(config):
        m, d = config.model, config.data
        return NetCfg(m.nf, m.ch_mult, m.num_res_blocks, m.attn_resolutions,
                      d.effective_image_size, m.input_channels, m.output_channels,
                      m.resamp_with_conv, m.conditional, d.centered, m.nonlinearity.lower())


def ddpm_module_list(cfg):
    """Module sequence of DDPM.all_modules (models/ddpm.py:96-147): list of
    (kind, idx, dict) in construction order - which is also execution order."""
    mods = []

    def add(kind, **kw):
        mods.append((kind, len(mods), kw))

    nf = cfg.nf
    if cfg.conditional:
        add('linear', cin=nf, cout=4 * nf)
        add('linear', cin=4 * nf, cout=4 * nf)
    add('conv3', cin=cfg.input_channels, cout=nf)
    nres = len(cfg.ch_mult)
    res = [cfg.image_size // (2 ** i) for i in range(nres)]
    hs_c = [nf]
    in_ch = nf
    for lvl in range(nres):
        for _ in range(cfg.num_res_blocks):
            out_ch = nf * cfg.ch_mult[lvl]
            add('res', cin=in_ch, cout=out_ch)
            in_ch = out_ch
            if res[lvl] in cfg.attn_resolutions:
                add('attn', ch=in_ch)
            hs_c.append(in_ch)
        if lvl != nres - 1:
            add('down', ch=in_ch)
            hs_c.append(in_ch)
    add('res', cin=in_ch, cout=in_ch)
    add('attn', ch=in_ch)
    add('res', cin=in_ch, cout=in_ch)
    for lvl in reversed(range(nres)):
        for _ in range(cfg.num_res_blocks + 1):
            out_ch = nf * cfg.ch_mult[lvl]
            add('res', cin=in_ch + hs_c.pop(), cout=out_ch)
            in_ch = out_ch
        if res[lvl] in cfg.attn_resolutions:
            add('attn', ch=in_ch)
        if lvl != 0:
            add('up', ch=in_ch)
    assert not hs_c
    add('gn', ch=in_ch)
    add('conv3', cin=in_ch, cout=cfg.output_channels)
    return mods


def ddpm_param_shapes(cfg):
    """{state_dict key: shape} for the DDPM family."""
    shapes = {}
    temb = 4 * cfg.nf
    for kind, i, kw in ddpm_module_list(cfg):
        p = 'all_modules.%d.' % i
        if kind == 'linear':
            shapes[p + 'weight'] = (kw['cout'], kw['cin'])
            shapes[p + 'bias'] = (kw['cout'],)
        elif kind == 'conv3':
            shapes[p + 'weight'] = (kw['cout'], kw['cin'], 3, 3)
            shapes[p + 'bias'] = (kw['cout'],)
        elif kind == 'gn':
            shapes[p + 'weight'] = (kw['ch'],)
            shapes[p + 'bias'] = (kw['ch'],)
        elif kind in ('down', 'up'):
            if cfg.resamp_with_conv:
                shapes[p + 'Conv_0.weight'] = (kw['ch'], kw['ch'], 3, 3)
                shapes[p + 'Conv_0.bias'] = (kw['ch'],)
        elif kind == 'attn':
            c = kw['ch']
            shapes[p + 'GroupNorm_0.weight'] = (c,)
            shapes[p + 'GroupNorm_0.bias'] = (c,)
            for j in range(4):
                shapes[p + 'NIN_%d.W' % j] = (c, c)
                shapes[p + 'NIN_%d.b' % j] = (c,)
        elif kind == 'res':
            ci, co = kw['cin'], kw['cout']
            shapes[p + 'GroupNorm_0.weight'] = (ci,)
            shapes[p + 'GroupNorm_0.bias'] = (ci,)
            shapes[p + 'Conv_0.weight'] = (co, ci, 3, 3)
            shapes[p + 'Conv_0.bias'] = (co,)
            if cfg.conditional:
                shapes[p + 'Dense_0.weight'] = (co, temb)
                shapes[p + 'Dense_0.bias'] = (co,)
            shapes[p + 'GroupNorm_1.weight'] = (co,)
            shapes[p + 'GroupNorm_1.bias'] = (co,)
            shapes[p + 'Conv_1.weight'] = (co, co, 3, 3)
            shapes[p + 'Conv_1.bias'] = (co,)
            if ci != co:
                shapes[p + 'NIN_0.W'] = (ci, co)
                shapes[p + 'NIN_0.b'] = (co,)
    return shapes


def synth_params(shapes, seed=0):
    """Deterministic NON-DEGENERATE synthetic weights: every >=2-D tensor is
    U(+-sqrt(3/fan_avg)) (the reference's ``default_init(1.0)``, models/layers.py:54-91), biases
    and GroupNorm affine parameters are perturbed so those code paths are exercised.
    numpy's legacy RandomState is version-stable, so fixtures need not store the weights."""
    out = {}
    for k in sorted(shapes):
        shp = shapes[k]
        rs = np.random.RandomState((seed * 1000003 + _stable_hash(k)) % (2 ** 31 - 1))
        if len(shp) >= 2:
            if k.endswith('.W'):  # NIN: [in, out]
                fan_in, fan_out = shp[0], shp[1]
            else:  # conv OIHW / linear [out,in]
                rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
                fan_in, fan_out = shp[1] * rf, shp[0] * rf
            lim = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
            v = rs.uniform(-lim, lim, size=shp)
        elif 'GroupNorm' in k and k.endswith('weight') or (k.count('.') == 2 and k.endswith('weight')):
            v = 1.0 + 0.1 * rs.standard_normal(shp)
        else:
            v = 0.05 * rs.standard_normal(shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def _stable_hash(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


# ------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------
def _act(name):
    return {'swish': F.silu, 'relu': F.relu, 'elu': F.elu,
            'lrelu': lambda v: F.leaky_relu(v, 0.2)}[name]


def timestep_embedding(t, dim, max_positions=10000):
    """models/layers.py:524-538."""
    half = dim // 2
    w = math.log(max_positions) / (half - 1)
    w = torch.exp(torch.arange(half, dtype=torch.float32) * -w)
    e = t.float()[:, None] * w[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1))
    return e


def nin(x, W, b):
    """models/layers.py:555-564: NHWC matmul [.., Cin] @ W[Cin, Cout] + b."""
    return (x.permute(0, 2, 3, 1) @ W + b).permute(0, 3, 1, 2)


def attn_block(p, pre, x, groups=32):
    """models/layers.py:567-590."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, p[pre + 'GroupNorm_0.weight'], p[pre + 'GroupNorm_0.bias'], eps=1e-6)
    q = nin(h, p[pre + 'NIN_0.W'], p[pre + 'NIN_0.b'])
    k = nin(h, p[pre + 'NIN_1.W'], p[pre + 'NIN_1.b'])
    v = nin(h, p[pre + 'NIN_2.W'], p[pre + 'NIN_2.b'])
    w = torch.einsum('bchw,bcij->bhwij', q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, W, H * W), dim=-1).reshape(B, H, W, H, W)
    h = torch.einsum('bhwij,bcij->bchw', w, v)
    h = nin(h, p[pre + 'NIN_3.W'], p[pre + 'NIN_3.b'])
    return x + h


def res_block(p, pre, x, temb, act, cout, groups=32):
    """ResnetBlockDDPM.forward in eval mode - dropout is identity (models/layers.py:658-675)."""
    cin = x.shape[1]
    h = act(F.group_norm(x, groups, p[pre + 'GroupNorm_0.weight'], p[pre + 'GroupNorm_0.bias'], eps=1e-6))
    h = F.conv2d(h, p[pre + 'Conv_0.weight'], p[pre + 'Conv_0.bias'], padding=1)
    if temb is not None:
        h = h + F.linear(act(temb), p[pre + 'Dense_0.weight'], p[pre + 'Dense_0.bias'])[:, :, None, None]
    h = act(F.group_norm(h, groups, p[pre + 'GroupNorm_1.weight'], p[pre + 'GroupNorm_1.bias'], eps=1e-6))
    h = F.conv2d(h, p[pre + 'Conv_1.weight'], p[pre + 'Conv_1.bias'], padding=1)
    if cin != cout:
        x = nin(x, p[pre + 'NIN_0.W'], p[pre + 'NIN_0.b'])
    return x + h


def downsample(p, pre, x, with_conv):
    """models/layers.py:619-629: pad (0,1,0,1) then 3x3 stride-2 conv, or 2x2 avg-pool."""
    if with_conv:
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), p[pre + 'Conv_0.weight'], p[pre + 'Conv_0.bias'], stride=2)
    return F.avg_pool2d(x, 2, 2)


def upsample(p, pre, x, with_conv):
    """models/layers.py:600-604: nearest x2 then 3x3 conv."""
    h = F.interpolate(x, scale_factor=2, mode='nearest')
    if with_conv:
        h = F.conv2d(h, p[pre + 'Conv_0.weight'], p[pre + 'Conv_0.bias'], padding=1)
    return h


def ddpm_forward(p, cfg, x, labels):
    """DDPM.forward (models/ddpm.py:149-213). x: [B, input_channels, H, W] NCHW fp32."""
    act = _act(cfg.act)
    mods = ddpm_module_list(cfg)
    it = iter(mods)

    def nxt(kind):
        k, i, kw = next(it)
        assert k == kind, (k, kind)
        return 'all_modules.%d.' % i, kw

    temb = None
    if cfg.conditional:
        pre, _ = nxt('linear')
        temb = F.linear(timestep_embedding(labels, cfg.nf), p[pre + 'weight'], p[pre + 'bias'])
        pre, _ = nxt('linear')
        temb = F.linear(act(temb), p[pre + 'weight'], p[pre + 'bias'])
    h = x if cfg.centered else 2 * x - 1.
    pre, _ = nxt('conv3')
    hs = [F.conv2d(h, p[pre + 'weight'], p[pre + 'bias'], padding=1)]
    nres = len(cfg.ch_mult)
    for lvl in range(nres):
        for _ in range(cfg.num_res_blocks):
            pre, kw = nxt('res')
            h = res_block(p, pre, hs[-1], temb, act, kw['cout'])
            if h.shape[-1] in cfg.attn_resolutions:
                pre, _ = nxt('attn')
                h = attn_block(p, pre, h)
            hs.append(h)
        if lvl != nres - 1:
            pre, _ = nxt('down')
            hs.append(downsample(p, pre, hs[-1], cfg.resamp_with_conv))
    h = hs[-1]
    pre, kw = nxt('res')
    h = res_block(p, pre, h, temb, act, kw['cout'])
    pre, _ = nxt('attn')
    h = attn_block(p, pre, h)
    pre, kw = nxt('res')
    h = res_block(p, pre, h, temb, act, kw['cout'])
    for lvl in reversed(range(nres)):
        for _ in range(cfg.num_res_blocks + 1):
            pre, kw = nxt('res')
            h = res_block(p, pre, torch.cat([h, hs.pop()], dim=1), temb, act, kw['cout'])
        if h.shape[-1] in cfg.attn_resolutions:
            pre, _ = nxt('attn')
            h = attn_block(p, pre, h)
        if lvl != 0:
            pre, _ = nxt('up')
            h = upsample(p, pre, h, cfg.resamp_with_conv)
    assert not hs
    pre, _ = nxt('gn')
    h = act(F.group_norm(h, 32, p[pre + 'weight'], p[pre + 'bias'], eps=1e-6))
    pre, _ = nxt('conv3')
    return F.conv2d(h, p[pre + 'weight'], p[pre + 'bias'], padding=1)


def paired_forward(p, cfg, x, y, labels, sr3):
    """DDPM_paired_SR3 / DDPM_paired forward (models/ddpm.py:275-298)."""
    out = ddpm_forward(p, cfg, torch.cat((x, y), dim=1), labels)
    if sr3:
        return out
    c = x.shape[1]
    return {'x': out[:, :c], 'y': out[:, c:]}


def upfirdn2d_ref(x, kernel, up=1, down=1, pad=(0, 0)):
    """CPU branch of the reference's upfirdn2d (op/upfirdn2d.py:161-202 ``upfirdn2d_native``):
    zero-stuff by `up`, pad/crop, correlate with the FLIPPED kernel, keep every `down`-th sample."""
    N, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    out = x.reshape(N * C, H, 1, W, 1)
    out = F.pad(out, [0, up - 1, 0, 0, 0, up - 1])
    out = out.reshape(N * C, 1, H * up, W * up)
    out = F.pad(out, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    out = out[:, :, max(-p0, 0):out.shape[2] - max(-p1, 0), max(-p0, 0):out.shape[3] - max(-p1, 0)]
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)
    out = F.conv2d(out, w)
    out = out[:, :, ::down, ::down]
    return out.reshape(N, C, out.shape[2], out.shape[3])


# ------------------------------------------------------------------------------------------
# NCSN++ (models/ncsnpp.py; blocks models/layerspp.py; FIR resampling models/up_or_down_sampling.py)
# ------------------------------------------------------------------------------------------
def _ncsnpp_groups(c):
    return min(c // 4, 32)                     # layerspp.py:67,219,231


def fir_kernel_2d(k, gain=1.0):
    """up_or_down_sampling.py:181-189 ``_setup_kernel`` (separable taps -> normalised 2-D kernel) times a gain."""
    k = np.asarray(k, dtype=np.float32)
    k = np.outer(k, k)
    k /= np.sum(k)
    return torch.tensor(k * gain)


def upsample_2d(x, k, factor=2):
    """up_or_down_sampling.py:196-226."""
    kern = fir_kernel_2d(k, float(factor ** 2))
    p = kern.shape[0] - factor
    return upfirdn2d_ref(x, kern, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k, factor=2):
    """up_or_down_sampling.py:229-257."""
    kern = fir_kernel_2d(k, 1.0)
    p = kern.shape[0] - factor
    return upfirdn2d_ref(x, kern, down=factor, pad=((p + 1) // 2, p // 2))


def naive_upsample_2d(x, factor=2):
    """up_or_down_sampling.py:59-63 (fir = False): nearest-neighbour."""
    N, C, H, W = x.shape
    return x.reshape(N, C, H, 1, W, 1).repeat(1, 1, 1, factor, 1, factor).reshape(N, C, H * factor, W * factor)


def naive_downsample_2d(x, factor=2):
    """up_or_down_sampling.py:66-69 (fir = False): mean over factor x factor blocks."""
    N, C, H, W = x.shape
    return torch.mean(x.reshape(N, C, H // factor, factor, W // factor, factor), dim=(3, 5))


def resample(x, fir_k, up):
    """FIR resampling (fir_k a tap tuple) or, for fir_k None, the naive forms - as ResnetBlockBigGANpp (layerspp.py:246-259),
    layerspp.Upsample / Downsample without conv (:110-123,149-163) select them by ``fir``."""
    if fir_k is None:
        return naive_upsample_2d(x) if up else naive_downsample_2d(x)
    return upsample_2d(x, fir_k) if up else downsample_2d(x, fir_k)


def pyramid_down_conv(p, pre, x, fir_k):
    """layerspp.Downsample(with_conv=True) (layerspp.py:130-165): fir -> up_or_down_sampling.Conv2d(down=True) =
    conv_downsample_2d (:155-178: FIR with pad ((p+1)//2, p//2), p = (taps - 2) + 2, then a VALID stride-2 3x3 conv) + bias;
    no fir -> pad (0,1,0,1) + stride-2 conv3x3."""
    if fir_k is None:
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), p[pre + 'Conv_0.weight'], p[pre + 'Conv_0.bias'], stride=2)
    kern = fir_kernel_2d(fir_k, 1.0)
    pp = (kern.shape[0] - 2) + 2
    z = upfirdn2d_ref(x, kern, pad=((pp + 1) // 2, pp // 2))
    return F.conv2d(z, p[pre + 'Conv2d_0.weight'], stride=2) + p[pre + 'Conv2d_0.bias'].reshape(1, -1, 1, 1)


def biggan_block(p, pre, x, temb, act, cin, cout, up, down, fir_k, skip_rescale):
    """ResnetBlockBigGANpp.forward in eval mode (layerspp.py:242-274); fir_k None = fir False."""
    h = act(F.group_norm(x, _ncsnpp_groups(cin), p[pre + 'GroupNorm_0.weight'], p[pre + 'GroupNorm_0.bias'], eps=1e-6))
    if up:
        h, x = resample(h, fir_k, True), resample(x, fir_k, True)
    elif down:
        h, x = resample(h, fir_k, False), resample(x, fir_k, False)
    h = F.conv2d(h, p[pre + 'Conv_0.weight'], p[pre + 'Conv_0.bias'], padding=1)
    if temb is not None:
        h = h + F.linear(act(temb), p[pre + 'Dense_0.weight'], p[pre + 'Dense_0.bias'])[:, :, None, None]
    h = act(F.group_norm(h, _ncsnpp_groups(cout), p[pre + 'GroupNorm_1.weight'], p[pre + 'GroupNorm_1.bias'], eps=1e-6))
    h = F.conv2d(h, p[pre + 'Conv_1.weight'], p[pre + 'Conv_1.bias'], padding=1)
    if cin != cout or up or down:
        x = F.conv2d(x, p[pre + 'Conv_2.weight'], p[pre + 'Conv_2.bias'])
    return (x + h) / np.sqrt(2.) if skip_rescale else x + h


def attn_block_pp(p, pre, x, skip_rescale):
    """AttnBlockpp.forward (layerspp.py:75-91): AttnBlock with GroupNorm(min(C/4, 32)) and the optional rescale."""
    C = x.shape[1]
    out = attn_block(p, pre, x, groups=_ncsnpp_groups(C))          # = x + h
    return out / np.sqrt(2.) if skip_rescale else out


def ncsnpp_forward(p, config, x, time_cond):
    """NCSNpp.forward (models/ncsnpp.py:238-388) for the option set the HIP adapter covers: biggan blocks, FIR or naive
    resampling, progressive in {none, output_skip}, progressive_input in {none, input_skip, residual}, combine 'sum'."""
    m, d = config.model, config.data
    act = _act(m.nonlinearity.lower())
    nf, ch_mult = m.nf, tuple(m.ch_mult)
    L = len(ch_mult)
    res = [d.effective_image_size // (2 ** i) for i in range(L)]
    fir_k, skip = (tuple(m.fir_kernel) if m.fir else None), bool(m.skip_rescale)
    prog, prog_in = m.progressive.lower(), m.progressive_input.lower()
    i = 0

    def pre():
        return 'all_modules.%d.' % i

    if m.embedding_type.lower() == 'fourier':
        xp = time_cond[:, None] * p[pre() + 'W'][None, :] * 2 * np.pi          # layerspp.py:39-41
        temb = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
        i += 1
    else:
        temb = timestep_embedding(time_cond, nf)
    if m.conditional:
        temb = F.linear(temb, p[pre() + 'weight'], p[pre() + 'bias'])
        i += 1
        temb = F.linear(act(temb), p[pre() + 'weight'], p[pre() + 'bias'])
        i += 1
    else:
        temb = None
    if not d.centered:
        x = 2 * x - 1.
    input_pyramid = x if prog_in != 'none' else None
    hs = [F.conv2d(x, p[pre() + 'weight'], p[pre() + 'bias'], padding=1)]
    i += 1
    in_ch = nf
    hs_c = [nf]
    for lv in range(L):
        for _ in range(m.num_res_blocks):
            out_ch = nf * ch_mult[lv]
            h = biggan_block(p, pre(), hs[-1], temb, act, in_ch, out_ch, False, False, fir_k, skip)
            i += 1
            in_ch = out_ch
            if h.shape[-1] in m.attn_resolutions:
                h = attn_block_pp(p, pre(), h, skip)
                i += 1
            hs.append(h)
            hs_c.append(in_ch)
        if lv != L - 1:
            h = biggan_block(p, pre(), hs[-1], temb, act, in_ch, in_ch, False, True, fir_k, skip)
            i += 1
            if prog_in == 'input_skip':
                input_pyramid = resample(input_pyramid, fir_k, False)
                h = F.conv2d(input_pyramid, p[pre() + 'Conv_0.weight'], p[pre() + 'Conv_0.bias']) + h   # Combine 'sum'
                i += 1
            elif prog_in == 'residual':                                        # ncsnpp.py:302-309
                input_pyramid = pyramid_down_conv(p, pre(), input_pyramid, fir_k)
                i += 1
                input_pyramid = (input_pyramid + h) / np.sqrt(2.) if skip else input_pyramid + h
                h = input_pyramid
            hs.append(h)
            hs_c.append(in_ch)
    h = hs[-1]
    h = biggan_block(p, pre(), h, temb, act, in_ch, in_ch, False, False, fir_k, skip)
    i += 1
    h = attn_block_pp(p, pre(), h, skip)
    i += 1
    h = biggan_block(p, pre(), h, temb, act, in_ch, in_ch, False, False, fir_k, skip)
    i += 1
    pyramid = None
    for lv in reversed(range(L)):
        for _ in range(m.num_res_blocks + 1):
            out_ch = nf * ch_mult[lv]
            h = biggan_block(p, pre(), torch.cat([h, hs.pop()], dim=1), temb, act, in_ch + hs_c.pop(), out_ch, False, False,
                             fir_k, skip)
            i += 1
            in_ch = out_ch
        if h.shape[-1] in m.attn_resolutions:
            h = attn_block_pp(p, pre(), h, skip)
            i += 1
        if prog == 'output_skip':
            ph = act(F.group_norm(h, _ncsnpp_groups(in_ch), p[pre() + 'weight'], p[pre() + 'bias'], eps=1e-6))
            i += 1
            ph = F.conv2d(ph, p[pre() + 'weight'], p[pre() + 'bias'], padding=1)
            i += 1
            pyramid = ph if pyramid is None else resample(pyramid, fir_k, True) + ph
        if lv != 0:
            h = biggan_block(p, pre(), h, temb, act, in_ch, in_ch, True, False, fir_k, skip)
            i += 1
    assert not hs
    if prog == 'output_skip':
        return pyramid
    h = act(F.group_norm(h, _ncsnpp_groups(in_ch), p[pre() + 'weight'], p[pre() + 'bias'], eps=1e-6))
    i += 1
    return F.conv2d(h, p[pre() + 'weight'], p[pre() + 'bias'], padding=1)


# ------------------------------------------------------------------------------------------
# VE SDE scalars (fp32, exactly the reference's expressions)
# ------------------------------------------------------------------------------------------
class VE:
    """sigma table + per-t scalars of (c)VESDE (sde_lib.py:290-418)."""

    def __init__(self, sigma_min, sigma_max, N=1000):
        self.sigma_min, self.sigma_max, self.N, self.T = sigma_min, sigma_max, N, 1
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    def std(self, t):
        smin = torch.tensor(self.sigma_min).type_as(t)
        smax = torch.tensor(self.sigma_max).type_as(t)
        return smin * (smax / smin) ** t

    def G(self, t):
        i = (t * (self.N - 1) / self.T).long()
        sig = self.discrete_sigmas[i]
        adj = torch.where(i == 0, torch.zeros_like(t), self.discrete_sigmas[i - 1])
        return torch.sqrt(sig ** 2 - adj ** 2)


def _b(v):
    return v[:, None, None, None]


def score_sr3(p, cfg, ve_x, x, y, t):
    """cVESDE continuous branch of get_score_fn (models/utils.py:210-215) + divide_by_sigmas."""
    labels = t * (ve_x.N - 1)
    out = paired_forward(p, cfg, x, y, labels, sr3=True)
    return out / _b(ve_x.std(t))


def score_paired_x(p, cfg, ve_x, ve_y, x, y, t):
    """dict-SDE branch (models/utils.py:173-180); conditional wrapper keeps ['x'] (:270-278)."""
    labels = t * (ve_x.N - 1)
    out = paired_forward(p, cfg, x, y, labels, sr3=False)
    return out['x'] / _b(ve_x.std(t))


class NoiseTape:
    """Serves pre-drawn standard-normal tensors in call order."""

    def __init__(self, tensors):
        self.t, self.i = list(tensors), 0

    def __call__(self, like):
        z = self.t[self.i]
        self.i += 1
        assert z.shape == like.shape
        return z


def langevin_update(score, x, z, snr, batch_reduce=None):
    """conditionalLangevinCorrector.update_fn, VE (alpha=1) (sampling/correctors.py:88-108).
    ``batch_reduce`` lets the multi-GPU "global-norm" mode all-reduce the two batch means."""
    g = torch.norm(score.reshape(score.shape[0], -1), dim=-1).mean()
    n = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
    if batch_reduce is not None:
        g, n = batch_reduce(g, n)
    step = (snr * n / g) ** 2 * 2 * torch.ones(x.shape[0])
    x_mean = x + _b(step) * score
    return x_mean + _b(torch.sqrt(step * 2)) * z, x_mean


def reverse_diffusion_update(score, x, z, G):
    """conditionalReverseDiffusionPredictor.update_fn with cVESDE.discretize
    (sampling/predictors.py:97-102; sde_lib.py:135-140,410-418): f=0, rev_f=-G^2*score."""
    rev_f = torch.zeros_like(x) - _b(G) ** 2 * score
    x_mean = x - rev_f
    return x_mean + _b(G) * z, x_mean


def pc_sample_conditional(p, cfg, y, noise, sigma_x, sigma_y=None, sr3=True, p_steps=1000,
                          snr=0.15, eps=1e-5, denoise=True, N=1000, record=None, max_steps=None):
    """Default (non-use_path) loop of get_pc_conditional_sampler (sampling/conditional.py:180-226).

    noise: NoiseTape; draw order = prior, then per step [z_y(corr)], z_corr, [z_y(pred)], z_pred
    (bracketed draws only for the dict-SDE estimators CMDE / VS-CMDE).
    sigma_x / sigma_y: (sigma_min, sigma_max) tuples. Corrector THEN predictor, c_steps=1."""
    ve_x = VE(*sigma_x, N=N)
    ve_y = VE(*sigma_y, N=N) if sigma_y is not None else None
    B = y.shape[0]
    xshape = (B, cfg.output_channels if sr3 else y.shape[1]) + tuple(y.shape[2:])
    x = noise(torch.empty(xshape)) * ve_x.sigma_max
    timesteps = torch.linspace(ve_x.T, eps, p_steps)
    x_mean = x
    for i in range(p_steps if max_steps is None else min(p_steps, max_steps)):   # max_steps: bounded bench sample
        vec_t = torch.ones(B) * timesteps[i]
        for phase in ('corrector', 'predictor'):
            if ve_y is not None:
                y_in = y + noise(y) * _b(ve_y.std(vec_t))
                s = score_paired_x(p, cfg, ve_x, ve_y, x, y_in, vec_t)
            else:
                s = score_sr3(p, cfg, ve_x, x, y, vec_t)
            z = noise(x)
            if phase == 'corrector':
                x, x_mean = langevin_update(s, x, z, snr)
            else:
                x, x_mean = reverse_diffusion_update(s, x, z, ve_x.G(vec_t))
        if record is not None:
            record.append(x.clone())
    return x_mean if denoise else x


def pc_sample_unconditional(score_fn, shape, noise, ve, p_steps=1000, snr=0.15, eps=1e-5, denoise=True):
    """get_pc_sampler loop (sampling/unconditional.py:194-226): LangevinCorrector then
    ReverseDiffusionPredictor, same update algebra as the conditional pair."""
    x = noise(torch.empty(shape)) * ve.sigma_max
    timesteps = torch.linspace(ve.T, eps, p_steps)
    x_mean = x
    for i in range(p_steps):
        vec_t = torch.ones(shape[0]) * timesteps[i]
        s = score_fn(x, vec_t)
        x, x_mean = langevin_update(s, x, noise(x), snr)
        s = score_fn(x, vec_t)
        x, x_mean = reverse_diffusion_update(s, x, noise(x), ve.G(vec_t))
    return x_mean if denoise else x


# ------------------------------------------------------------------------------------------
# denoising score-matching loss (training objective), VE SDEs
# ------------------------------------------------------------------------------------------
def _g2(ve, t):
    """diffusion^2 of (c)VESDE.sde (sde_lib.py:310-314,383-388): (sigma(t) sqrt(2 ln(smax/smin)))^2"""
    return (ve.std(t) * torch.sqrt(torch.tensor(2 * (np.log(ve.sigma_max) - np.log(ve.sigma_min))).type_as(t))) ** 2


def dsm_loss(p, cfg, name, ve_x, ve_y, x, y, t, z_x, z_y=None, reduce_mean=True, likelihood_weighting=True):
    """get_general_sde_loss_fn (losses.py:99-232) for the VE SDEs with supplied t and noise (dropout off):
    ``name`` 'ddpm' = unconditional branch (:208-232, score_fn models/utils.py:246-253: label sigma(t)),
    'ddpm_paired_SR3' = SR3 branch (:185-205), 'ddpm_paired' = two-SDE branch (:120-146)."""
    red = (lambda v: v.mean(dim=-1)) if reduce_mean else (lambda v: 0.5 * v.sum(dim=-1))
    std_x = ve_x.std(t)
    xt = x + _b(std_x) * z_x
    if name == 'ddpm_paired':
        std_y = ve_y.std(t)
        yt = y + _b(std_y) * z_y
        out = paired_forward(p, cfg, xt, yt, t * (ve_x.N - 1), sr3=False)
        sx, sy = out['x'] / _b(std_x), out['y'] / _b(std_y)
        lx = torch.square(sx + z_x / _b(std_x)) * _b(_g2(ve_x, t))
        ly = torch.square(sy + z_y / _b(std_y)) * _b(_g2(ve_y, t))
        losses = red(torch.cat([lx.reshape(lx.shape[0], -1), ly.reshape(ly.shape[0], -1)], dim=-1))
        return losses.mean()
    if name == 'ddpm':
        score = ddpm_forward(p, cfg, xt, std_x) / _b(std_x)
    else:
        score = paired_forward(p, cfg, xt, y, t * (ve_x.N - 1), sr3=True) / _b(std_x)
    if likelihood_weighting:
        losses = red(torch.square(score + z_x / _b(std_x)).reshape(x.shape[0], -1)) * _g2(ve_x, t)
    else:
        losses = red(torch.square(score * _b(std_x) + z_x).reshape(x.shape[0], -1))
    return losses.mean()


def pf_ode_sample(score_fn, ve, shape, z, eps=1e-5, rtol=1e-5, atol=1e-5, denoise=True):
    """Probability-flow ODE sampler (sampling/unconditional.py:93-158) for a VE SDE: scipy RK45 on dx/dt = -g(t)^2 score / 2
    (sde_lib.py:123-133 with f = 0), then one noise-free reverse-diffusion step at t = eps (predictors.py:84-89).
    ``score_fn(x, t) -> score``.  Returns (x, nfe)."""
    from scipy import integrate

    def ode_func(t, xf):
        x = torch.from_numpy(xf.reshape(shape)).type(torch.float32)
        vt = torch.ones(shape[0]) * t
        drift = -_b(_g2(ve, vt)) * score_fn(x, vt) * 0.5
        return drift.detach().numpy().reshape((-1,))

    sol = integrate.solve_ivp(ode_func, (ve.T, eps), z.detach().numpy().reshape((-1,)), rtol=rtol, atol=atol, method='RK45')
    x = torch.tensor(sol.y[:, -1]).reshape(shape).type(torch.float32)
    if denoise:
        vt = torch.ones(shape[0]) * eps
        x = x + _b(ve.G(vt)) ** 2 * score_fn(x, vt)
    return x, sol.nfev


def pc_inpaint_unconditional(score_fn, data, mask, noise, ve, snr=0.15, eps=1e-5, denoise=True):
    """get_pc_inpainter (sampling/unconditional.py:230-345) for a VE SDE with the (reverse diffusion, Langevin) pair: N = ve.N steps;
    after each update x = x(1 - mask) + (data + sigma(t) z) mask, x_mean = x(1 - mask) + data mask.  ``noise(like)`` serves the
    draws in the reference's order: prior; per step corrector z, blend z, predictor z, blend z."""
    x = data * mask + noise(data) * ve.sigma_max * (1. - mask)
    timesteps = torch.linspace(ve.T, eps, ve.N)
    x_mean = x
    for i in range(ve.N):
        vec_t = torch.ones(data.shape[0]) * timesteps[i]
        for upd in ('c', 'p'):
            s = score_fn(x, vec_t)
            if upd == 'c':
                x, x_mean = langevin_update(s, x, noise(x), snr)
            else:
                x, x_mean = reverse_diffusion_update(s, x, noise(x), ve.G(vec_t))
            masked = data + noise(x) * _b(ve.std(vec_t))
            x = x * (1. - mask) + masked * mask
            x_mean = x * (1. - mask) + data * mask
    return x_mean if denoise else x
