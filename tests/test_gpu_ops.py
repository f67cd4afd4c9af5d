"""-m gpu: per-operator parity of the HIP kernels (through the C ABI) against plain fp32 PyTorch
on the CPU.  Tolerances: fp32 round-off of a different summation order; stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import score_oracle as so  # noqa: E402


def dev():
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g) * scale


CONV_CASES = [
    # B, Cin, Cout, H, ksize, stride, up2
    (2, 8, 32, 20, 3, 1, False),
    (2, 32, 64, 10, 3, 1, False),
    (1, 96, 96, 16, 3, 1, False),
    (2, 6, 32, 20, 3, 1, False),      # Cin padded to 8
    (2, 96, 3, 20, 3, 1, False),      # Cout padded to 32
    (3, 32, 32, 10, 3, 2, False),     # Downsample: pad (0,1,0,1), stride 2 -> 5x5
    (2, 64, 64, 16, 3, 2, False),
    (3, 32, 32, 5, 3, 1, True),       # Upsample: nearest x2 then conv -> 10x10
    (2, 64, 96, 8, 3, 1, True),
    (2, 64, 96, 5, 1, 1, False),      # NIN / 1x1
    (3, 192, 96, 20, 1, 1, False),    # NIN shortcut of an up-path block (fp16 modes: pointwise kernel, 64-pixel tiles)
    (2, 40, 200, 7, 1, 1, False),     # 1x1 with Cin padded to 32, three cout groups, ragged pixel tail
    (4, 288, 288, 5, 3, 1, False),    # odd 5x5 level, tile straddles images
    (1, 192, 96, 40, 3, 1, False),
    (1, 96, 96, 160, 3, 1, False),    # full-resolution layer
]


@pytest.mark.parametrize('B,Cin,Cout,H,ks,stride,up2', CONV_CASES)
def test_conv2d(B, Cin, Cout, H, ks, stride, up2):
    from conditional_score_diffusion_amd import ops
    x = rnd(B, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2, scale=(1.0 / (Cin * ks * ks)) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    if up2:
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), w, b, padding=ks // 2)
    elif stride == 2:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    else:
        ref = F.conv2d(x, w, b, padding=ks // 2)
    out = ops.conv2d(x.to(dev()), w.to(dev()), b.to(dev()), stride=stride, downsample_pad=(stride == 2), up2=up2)
    assert out.shape == ref.shape
    assert rel(out, ref) < 2e-6 * max(1, (Cin * ks * ks) ** 0.5 / 8)   # fp32 dot of length K


F16_CASES = list(CONV_CASES)      # 3x3: stride 1, stride-2 Downsample, x2-upsample fused; 1x1: pointwise kernel


@pytest.mark.parametrize('B,Cin,Cout,H,ks,stride,up2', F16_CASES)
@pytest.mark.parametrize('precision,tol', [('fp16x3', 2e-6), ('fp16', 2e-3)])
def test_conv2d_fp16_mfma(B, Cin, Cout, H, ks, stride, up2, precision, tol):
    """split-fp16 (3 MFMAs) must be fp32-class; plain fp16 is bounded by the operand rounding (2^-11)"""
    from conditional_score_diffusion_amd import ops
    x = rnd(B, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, ks, ks, seed=2, scale=(1.0 / (Cin * ks * ks)) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if up2 else x
    if stride == 2:
        ref = F.conv2d(F.pad(xin.double(), (0, 1, 0, 1)), w.double(), b.double(), stride=2)
    else:
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=ks // 2)
    out = ops.conv2d(x.to(dev()), w.to(dev()), b.to(dev()), stride=stride, downsample_pad=(stride == 2), up2=up2,
                     precision=precision)
    assert rel(out, ref) < tol * max(1, (Cin * 9) ** 0.5 / 8)


@pytest.mark.parametrize('B,C,H,G,act', [(2, 32, 20, 32, 'swish'), (3, 96, 10, 32, 'none'), (2, 288, 5, 32, 'swish'),
                                         (1, 96, 160, 32, 'swish'), (2, 64, 8, 16, 'relu')])
def test_groupnorm_act(B, C, H, G, act):
    from conditional_score_diffusion_amd import ops
    x = rnd(B, C, H, H, seed=4) * 3 + 0.7
    ga, be = 1 + 0.1 * rnd(C, seed=5), 0.1 * rnd(C, seed=6)
    ref = F.group_norm(x, G, ga, be, eps=1e-6)
    ref = {'swish': F.silu, 'none': lambda v: v, 'relu': F.relu}[act](ref)
    out = ops.groupnorm_act(x.to(dev()), ga.to(dev()), be.to(dev()), groups=G, eps=1e-6, act=act)
    assert rel(out, ref) < 5e-6


@pytest.mark.parametrize('B,C,H', [(2, 64, 5), (2, 32, 10), (1, 192, 20), (2, 96, 16), (3, 288, 10), (2, 288, 5)])
def test_attention(B, C, H):
    from conditional_score_diffusion_amd import ops
    q, k, v = rnd(B, C, H, H, seed=7), rnd(B, C, H, H, seed=8), rnd(B, C, H, H, seed=9)
    w = torch.einsum('bchw,bcij->bhwij', q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, H, H * H), dim=-1).reshape(B, H, H, H, H)
    ref = torch.einsum('bhwij,bcij->bchw', w, v)
    out = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()))
    assert rel(out, ref) < 1e-5


def test_attention_peaked_softmax():
    """forces large score spread so the online-softmax rescale branch matters (guide rule 26)"""
    from conditional_score_diffusion_amd import ops
    B, C, H = 1, 64, 20
    q, k, v = rnd(B, C, H, H, seed=1) * 4, rnd(B, C, H, H, seed=2) * 4, rnd(B, C, H, H, seed=3)
    k[:, :, 19, 19] = q[:, :, 0, 0] * 3          # a late key that dominates query 0
    w = torch.einsum('bchw,bcij->bhwij', q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, H, H * H).double(), dim=-1).reshape(B, H, H, H, H)
    ref = torch.einsum('bhwij,bcij->bchw', w, v.double())
    out = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()))
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize('B,K,N,act', [(64, 384, 6336, 'swish'), (1, 96, 384, 'none'), (70, 300, 37, 'swish'), (3, 258, 16, 'none'),
                                       (130, 7, 5, 'swish'), (5, 512, 130, 'relu')])
def test_linear_ragged_shapes(B, K, N, act):
    """csd_linear (lane = sample, LDS-staged operands): batches over 64 (two sample blocks), K not a multiple of 4 (dword path) and of the
    256-wide chunk, N not a multiple of the 16 features of a workgroup, the input activation - against fp64"""
    from conditional_score_diffusion_amd import ops
    x, w, b = rnd(B, K, seed=K + B), rnd(N, K, seed=N) * (K ** -0.5), rnd(N, seed=3)
    xa = {'swish': F.silu, 'relu': F.relu, 'none': (lambda t: t)}[act](x.double())
    ref = xa @ w.double().t() + b.double()
    out = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), act_in=act)
    assert rel(out, ref) < 1e-5
    out_nb = ops.linear(x.to(dev()), w.to(dev()), None, act_in=act)
    assert rel(out_nb, ref - b.double()) < 1e-5


def test_linear_same_bits_for_any_batch():
    """one sequential fp32 sum per output: row 17 of a B = 64 call equals the B = 1 call bit for bit"""
    from conditional_score_diffusion_amd import ops
    x, w, b = rnd(64, 384, seed=1), rnd(200, 384, seed=2) * 0.05, rnd(200, seed=3)
    full = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), act_in='swish')
    one = ops.linear(x[17:18].contiguous().to(dev()), w.to(dev()), b.to(dev()), act_in='swish')
    assert torch.equal(full[17:18], one)


@pytest.mark.parametrize('precision,tol', [('fp16x3', 1e-5), ('fp16f8', 1e-5), ('fp16', 3e-3), ('fp32', 1e-5)])
@pytest.mark.parametrize('B,L,C', [(3, 400, 192), (2, 100, 288), (2, 25, 288), (2, 256, 128), (1, 64, 256), (2, 37, 32), (1, 129, 96),
                                   (1, 160, 64)])
def test_attention_core_in_every_precision_mode(B, L, C, precision, tol):
    """csd_attention_nhwc_prec (what the planned network runs in each arithmetic mode) on the packed [B, L, 3C] tensor: ragged key
    tiles (L % 32 != 0), every instantiated width, a dominating late key (online-softmax rescale)"""
    from conditional_score_diffusion_amd import _lib
    qkv = rnd(B, L, 3 * C, seed=L + C) * 2
    qkv[:, L - 1, C:2 * C] = qkv[:, 0, :C] * 3          # the last key dominates query 0
    q, k, v = qkv[..., :C].double(), qkv[..., C:2 * C].double(), qkv[..., 2 * C:].double()
    w = torch.softmax(torch.einsum('bqc,bkc->bqk', q, k) * (int(C) ** (-0.5)), dim=-1)
    ref = torch.einsum('bqk,bkc->bqc', w, v)
    g = qkv.to(dev())
    out = torch.empty(B, L, C, device=dev())
    _lib.check(_lib.lib().csd_attention_nhwc_prec(_lib.ptr(g), _lib.ptr(out), B, L, C, _lib.PREC_IDS[precision],
                                                  _lib.current_stream(g.device)), 'attention_nhwc_prec')
    assert rel(out, ref) < tol


@pytest.mark.parametrize('up,down,pad', [(2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (1, 2)), (2, 2, (0, 0))])
def test_upfirdn2d(up, down, pad):
    from conditional_score_diffusion_amd import ops
    x = rnd(2, 5, 12, 12, seed=11)
    k1 = torch.tensor([1., 3., 3., 1.])
    k = torch.outer(k1, k1)
    k = k / k.sum() * (up ** 2)
    ref = so.upfirdn2d_ref(x, k, up, down, pad)
    out = ops.upfirdn2d(x.to(dev()), k.to(dev()), up, down, pad)
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-6


def test_fused_leaky_relu_and_up():
    from conditional_score_diffusion_amd import ops
    x, b = rnd(2, 6, 7, 7, seed=12), rnd(6, seed=13)
    ref = F.leaky_relu(x + b.view(1, -1, 1, 1), 0.2) * (2 ** 0.5)
    assert rel(ops.fused_leaky_relu(x.to(dev()), b.to(dev())), ref) < 1e-6
    ref = F.interpolate(x, scale_factor=2, mode='nearest')
    assert torch.equal(ops.nearest_up2(x.to(dev())).cpu(), ref)


def test_timestep_embedding():
    from conditional_score_diffusion_amd import ops
    t = torch.tensor([999.0, 978.6124, 500.25, 0.00999, 277.128])
    ref = so.timestep_embedding(t, 96)
    out = ops.timestep_embedding(t.to(dev()), 96).cpu()
    assert float((out - ref).abs().max()) < 1.5e-4   # 1 ulp of the frequency * t=999 moves sin by ~6e-5
    assert float((out - ref).abs().mean()) < 1e-5


def test_randn_statistics_and_determinism():
    from conditional_score_diffusion_amd import ops
    a = ops.randn((4, 3, 160, 160), 42, 7, dev())
    b = ops.randn((4, 3, 160, 160), 42, 7, dev())
    c = ops.randn((4, 3, 160, 160), 42, 8, dev())
    assert torch.equal(a, b) and not torch.equal(a, c)
    a = a.cpu().double()
    n = a.numel()
    assert abs(float(a.mean())) < 5 / n ** 0.5
    assert abs(float(a.var()) - 1) < 0.01
    assert abs(float((a ** 4).mean()) - 3) < 0.1
    assert torch.isfinite(a).all()


def test_update_kernels():
    from conditional_score_diffusion_amd import ops
    B = 3
    x, net, z = rnd(B, 3, 20, 20, seed=1) * 50, rnd(B, 3, 20, 20, seed=2), rnd(B, 3, 20, 20, seed=3)
    std, snr, G = 37.5, 0.15, 4.2
    xr, xmr = so.langevin_update(net / std, x, z, snr)
    xo, xmo = ops.langevin_step(x.to(dev()).clone(), net.to(dev()), z.to(dev()), std, snr)
    assert rel(xo, xr) < 1e-6 and rel(xmo, xmr) < 1e-6
    xr, xmr = so.reverse_diffusion_update(net / std, x, z, torch.full((B,), G))
    xo, xmo = ops.reverse_diffusion_step(x.to(dev()).clone(), net.to(dev()), z.to(dev()), std, G)
    assert rel(xo, xr) < 1e-6 and rel(xmo, xmr) < 1e-6
    sc = torch.tensor([2.0, 0.5, 277.0])
    assert rel(ops.scale_rows(x.to(dev()), sc.to(dev()), divide=True), x / sc.view(-1, 1, 1, 1)) < 1e-7


def _upfirdn2d_native(x, kernel, up, down, pad):
    """CPU restatement of op/upfirdn2d.py:161-202 (zero-stuff, pad / crop, correlate with the flipped taps, decimate)."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    out = x.reshape(-1, h, 1, w, 1, 1)
    out = F.pad(out, [0, 0, 0, up - 1, 0, 0, 0, up - 1]).reshape(-1, h * up, w * up, 1)
    p0, p1 = pad
    out = F.pad(out, [0, 0, max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    out = out[:, max(-p0, 0):out.shape[1] - max(-p1, 0), max(-p0, 0):out.shape[2] - max(-p1, 0), :]
    out = out.permute(0, 3, 1, 2).reshape(-1, 1, h * up + p0 + p1, w * up + p0 + p1)
    out = F.conv2d(out, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw))
    out = out[:, :, ::down, ::down]
    return out.reshape(n, c, out.shape[2], out.shape[3])


@pytest.mark.parametrize('up,down,pad,hw', [(2, 1, (2, 1), 9), (1, 2, (1, 1), 12), (1, 1, (1, 2), 7), (2, 1, (0, 0), 6),
                                            (1, 2, (2, 2), 11), (2, 2, (1, 1), 8)])
def test_upfirdn2d_backward_matches_autograd_of_native(up, down, pad, hw):
    """a12: backward of upfirdn2d = the same op with the flipped kernel, swapped factors and g_pad
    (op/upfirdn2d.py:20-85,108-116); checked against torch autograd through the native CPU formulation,
    and the double backward against the forward formula."""
    from conditional_score_diffusion_amd import ops
    k1 = torch.tensor([1., 3., 3., 1.])
    kern = torch.outer(k1, k1)
    kern = kern / kern.sum() * (up ** 2)
    x = rnd(2, 3, hw, hw + 1, seed=5)
    xc = x.clone().requires_grad_(True)
    ref = _upfirdn2d_native(xc, kern, up, down, pad)
    go = rnd(*ref.shape, seed=6)
    ref.backward(go)
    xg = x.to(dev()).requires_grad_(True)
    out = ops.upfirdn2d(xg, kern.to(dev()), up=up, down=down, pad=pad)
    assert rel(out.detach(), ref.detach()) < 1e-6
    go_d = go.to(dev()).requires_grad_(True)
    gi, = torch.autograd.grad(out, xg, go_d, create_graph=True)
    assert rel(gi.detach(), xc.grad) < 1e-6
    # double backward (op/upfirdn2d.py:62-85): d<gi, v>/d(go) = upfirdn2d(v) with the forward parameters
    v = rnd(*x.shape, seed=7)
    gg, = torch.autograd.grad(gi, go_d, v.to(dev()))
    assert rel(gg, _upfirdn2d_native(v, kern, up, down, pad)) < 1e-6


@pytest.mark.gpu
def test_op_package_fused_leaky_relu_forward_backward_double_backward():
    """conditional_score_diffusion_amd.op (the reference's op/ package: upfirdn2d, fused_leaky_relu, FusedLeakyReLU): forward, the
    gradients w.r.t. input and bias, and the double backward, against torch autograd of lrelu(x + b) * scale."""
    from conditional_score_diffusion_amd import op
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(19)
    x = torch.from_numpy(rs.standard_normal((3, 8, 5, 6)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(8).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((3, 8, 5, 6)).astype(np.float32))
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(xr + br.view(1, -1, 1, 1), 0.1) * 1.7
    gx, gb = torch.autograd.grad(ref, (xr, br), g, create_graph=True)
    (gx * g).sum().backward()                                   # d/d(g)... second-order path w.r.t. nothing here; checks create_graph works
    xd, bd = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    out = op.fused_leaky_relu(xd, bd, negative_slope=0.1, scale=1.7)
    assert (out.cpu() - ref.detach()).abs().max() < 1e-6
    gd = g.to(dev).requires_grad_(True)
    hx, hb = torch.autograd.grad(out, (xd, bd), gd, create_graph=True)
    assert (hx.cpu() - gx.detach()).abs().max() < 1e-6
    assert (hb.cpu() - gb.detach()).abs().max() < 1e-4
    # double backward: the input gradient is linear in grad_output, its derivative w.r.t. grad_output applied to a vector v
    v = torch.from_numpy(rs.standard_normal((3, 8, 5, 6)).astype(np.float32))
    (hx * v.to(dev)).sum().backward()
    gr = g.clone().requires_grad_(True)
    xr2 = x.clone()
    ref_gx = torch.autograd.grad(torch.nn.functional.leaky_relu(xr2.requires_grad_(True) + b.view(1, -1, 1, 1), 0.1) * 1.7, xr2, gr,
                                 create_graph=True)[0]
    (ref_gx * v).sum().backward()
    assert (gd.grad.cpu() - gr.grad).abs().max() < 1e-6
    mod = op.FusedLeakyReLU(8).to(dev)
    assert mod(xd.detach()).shape == x.shape
    k = torch.tensor([[1., 3., 3., 1.]]).T @ torch.tensor([[1., 3., 3., 1.]]) / 64
    assert op.upfirdn2d(xd.detach(), k.to(dev), up=2, pad=(2, 1)).shape == (3, 8, 10, 12)


@pytest.mark.gpu
def test_up_or_down_sampling_module_vs_oracle():
    """models/up_or_down_sampling.py by name: upsample_2d / downsample_2d / naive_* / conv_downsample_2d against the oracle's
    restatements (upfirdn2d_ref-based, pinned to the reference's module outputs by test_oracle_golden)."""
    import score_oracle as so
    from conditional_score_diffusion_amd.models import up_or_down_sampling as U
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(23)
    x = torch.from_numpy(rs.standard_normal((2, 6, 10, 10)).astype(np.float32))
    xd = x.to(dev)
    k = (1, 3, 3, 1)
    for got, ref in ((U.upsample_2d(xd, k), so.upsample_2d(x, k)), (U.downsample_2d(xd, k), so.downsample_2d(x, k)),
                     (U.naive_upsample_2d(xd), so.naive_upsample_2d(x)), (U.naive_downsample_2d(xd), so.naive_downsample_2d(x)),
                     (U.upsample_2d(xd), so.naive_upsample_2d(x)), (U.downsample_2d(xd), so.naive_downsample_2d(x))):
        assert got.shape == ref.shape and (got.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
    w = torch.from_numpy((rs.standard_normal((8, 6, 3, 3)) * 0.2).astype(np.float32))
    kern = so.fir_kernel_2d(k, 1.0)
    ref = torch.nn.functional.conv2d(so.upfirdn2d_ref(x, kern, pad=(2, 2)), w, stride=2)
    got = U.conv_downsample_2d(xd, w.to(dev), k)
    assert got.shape == ref.shape and (got.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


def _block_case(rs, B, C0, C1, Cout, H, W, norm, res, precision):
    """one csd_conv3x3_block call on random data against fp64 torch: (relative max error, relative error of the tile partials)"""
    from conditional_score_diffusion_amd import ops
    Cin = C0 + C1
    x = torch.from_numpy(rs.standard_normal((B, H, W, Cin)).astype(np.float32)) * 1.5 + 0.2
    w = torch.from_numpy(rs.standard_normal((Cout, Cin, 3, 3)).astype(np.float32)) / (3.0 * Cin ** 0.5)
    bias = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, (B, Cin)).astype(np.float32)) if norm else None
    sh = torch.from_numpy((rs.standard_normal((B, Cin)) * 0.5).astype(np.float32)) if norm else None
    rv = torch.from_numpy(rs.standard_normal((B, H, W, Cout)).astype(np.float32)) * 2.0 if res else None
    d = dev()
    opt = lambda t: None if t is None else t.to(d)      # noqa: E731
    y, stats = ops.conv3x3_block(x[..., :C0].contiguous().to(d), w.to(d), bias.to(d), x1=x[..., C0:].contiguous().to(d) if C1 else None,
                                 nscale=opt(sc), nshift=opt(sh), res=opt(rv), out_scale=0.75, precision=precision, want_stats=True)
    xd = x.double()
    if norm:
        xd = torch.nn.functional.silu(xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :])
    ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if rv is not None:
        ref = ref + rv.double()
    ref = ref * 0.75
    yc = y.cpu().double()
    err = (yc - ref).abs().max().item() / ref.abs().max().item()
    yt = yc.reshape(B, H // 16, 16, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(-1, 256, Cout)
    serr = (stats.cpu()[:, :, 0] - yt.sum(1)).abs().max().item() / yt.abs().sum(1).max().item()
    return err, serr


def test_conv3x3_block_shape_sweep_fp16x3():
    """conv_xk.hip (the persistent software-pipelined form: every fp16x3 layer with an even number >= 4 of 16-channel stages) over a
    seeded sweep of shapes: 96- and 64-cout groups, one and two sources, 1 .. 23 tiles per workgroup walk (grids smaller than,
    equal to and larger than the CU count, tile counts that do not divide by the 8 XCDs), with and without the GroupNorm prologue and
    the residual; the layers it does not cover (odd stage counts, two stages) ride along on conv_ff"""
    rs = np.random.RandomState(2024)
    worst = 0.0
    shapes = [(1, 16, 16), (1, 16, 32), (3, 16, 16), (1, 48, 16), (2, 32, 48), (5, 16, 32), (1, 80, 80), (7, 32, 32), (2, 64, 96)]
    chans = [(64, 0, 64), (96, 0, 96), (64, 64, 128), (96, 96, 96), (128, 0, 192), (32, 96, 64), (160, 32, 288), (128, 128, 256),
             (80, 0, 96), (32, 0, 64), (48, 16, 192)]
    n = 0
    for (B, H, W) in shapes:
        for k in rs.choice(len(chans), size=4, replace=False):
            C0, C1, Cout = chans[k]
            norm, res = bool(rs.randint(2)), bool(rs.randint(2))
            err, serr = _block_case(rs, B, C0, C1, Cout, H, W, norm, res, 'fp16x3')
            assert err < 3e-6 and serr < 1e-5, (B, H, W, C0, C1, Cout, norm, res, err, serr)
            worst = max(worst, err)
            n += 1
    assert n == 36 and worst > 0


def test_conv3x3_block_ragged_tiles_fp16x3():
    """maps that 16 does not divide (the 40^2 level of the 160^2 networks: 8 | H, W) run on ragged 16 x 16 tiles in conv_xk.hip: lanes whose
    4 x 8-pixel block lies outside the image drop their stores, residual loads and statistics; the patch pads with zeros as at any
    border.  Output and the per-tile GroupNorm partials (valid pixels only) against fp64 torch"""
    from conditional_score_diffusion_amd import ops
    rs = np.random.RandomState(40)
    d = dev()
    for (B, H, W, C0, C1, Cout, norm, res) in ((2, 40, 40, 96, 0, 96, True, False), (3, 40, 40, 192, 0, 192, True, True),
                                               (1, 40, 56, 96, 96, 192, True, False), (2, 56, 40, 64, 0, 96, False, True),
                                               (1, 40, 40, 192, 96, 192, True, True)):
        Cin = C0 + C1
        x = torch.from_numpy(rs.randn(B, H, W, Cin).astype(np.float32) * 2.0 + 0.3)
        w = torch.from_numpy((rs.randn(Cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)).astype(np.float32))
        bias = torch.from_numpy(rs.randn(Cout).astype(np.float32))
        sc = torch.from_numpy((rs.rand(B, Cin) + 0.5).astype(np.float32)) if norm else None
        sh = torch.from_numpy((rs.randn(B, Cin) * 0.5).astype(np.float32)) if norm else None
        rv = torch.from_numpy((rs.randn(B, H, W, Cout) * 3.0).astype(np.float32)) if res else None
        opt = lambda t: None if t is None else t.to(d)      # noqa: E731
        x0 = x[..., :C0].contiguous().to(d)
        x1 = x[..., C0:].contiguous().to(d) if C1 else None
        y, stats = ops.conv3x3_block(x0, w.to(d), bias.to(d), x1=x1, nscale=opt(sc), nshift=opt(sh), res=opt(rv), out_scale=0.75,
                                     precision='fp16x3', want_stats=True)
        xd = x.double()
        if norm:
            xd = torch.nn.functional.silu(xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :])
        ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        if rv is not None:
            ref = ref + rv.double()
        ref = ref * 0.75
        yc = y.cpu().double()
        err = (yc - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-6, (B, H, W, C0, C1, Cout, err)
        ty, tx = (H + 15) // 16, (W + 15) // 16
        st = stats.cpu().reshape(B, ty, tx, Cout, 2)
        worst = 0.0
        for i in range(ty):
            for j in range(tx):
                blk = yc[:, 16 * i:16 * i + 16, 16 * j:16 * j + 16, :]
                worst = max(worst, (st[:, i, j, :, 0] - blk.sum((1, 2))).abs().max().item(), (st[:, i, j, :, 1] - (blk * blk).sum((1, 2))).abs().max().item() * 0.1)
        assert worst < 1e-5 * float((yc * yc).sum((1, 2)).max()), (B, H, W, worst)


def test_conv3x3_block_rejects_maps_outside_the_ragged_domain():
    """ragged tiles exist for the fp16x3 Winograd layers with 96-cout groups, 8 | H, W and tiles >= 65 % full: anything else that 16 does
    not divide fails loudly instead of running a kernel that would write outside the image"""
    from conditional_score_diffusion_amd import ops
    d = dev()
    for (H, W, Cin, Cout, prec) in ((24, 24, 96, 96, 'fp16x3'),      # 8 | 24, but 56 % full
                                    (40, 40, 128, 128, 'fp16x3'),    # 64-cout groups
                                    (40, 40, 96, 96, 'fp16f8'),      # the fp8-correction form tiles by 16 only
                                    (36, 40, 96, 96, 'fp16x3')):     # 36 is not a multiple of 8
        x = torch.randn(1, H, W, Cin, device=d)
        w = torch.randn(Cout, Cin, 3, 3, device=d) * 0.05
        b = torch.zeros(Cout, device=d)
        with pytest.raises(RuntimeError):
            ops.conv3x3_block(x, w, b, precision=prec)


def test_conv3x3_block_is_bitwise_repeatable_under_load():
    """conv_xk.hip issues its matrix instructions as asm statements (no compiler-inserted wait states): 30 launches of a chip-filling
    layer (8 x 160^2, 96 -> 96 and 64 + 64 -> 128, GroupNorm prologue + residual; several tiles per workgroup, hot chip) must be
    bit-identical - a timing-dependent read of a matrix result would show up here"""
    from conditional_score_diffusion_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(5)
    for (C0, C1, Cout) in ((96, 0, 96), (64, 64, 128)):
        B, H = 8, 160
        Cin = C0 + C1
        x = torch.randn(B, H, H, Cin, generator=g).to(d)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(d)
        bias = torch.randn(Cout, generator=g).to(d)
        sc, sh = (torch.rand(B, Cin, generator=g) + 0.5).to(d), torch.randn(B, Cin, generator=g).to(d)
        rv = torch.randn(B, H, H, Cout, generator=g).to(d)
        x0, x1 = x[..., :C0].contiguous(), (x[..., C0:].contiguous() if C1 else None)
        first = None
        for _ in range(30):
            y, st = ops.conv3x3_block(x0, w, bias, x1=x1, nscale=sc, nshift=sh, res=rv, precision='fp16x3', want_stats=True)
            if first is None:
                first = (y.clone(), st.clone())
            else:
                assert torch.equal(y, first[0]) and torch.equal(st, first[1])


@pytest.mark.parametrize('precision,tol', [('fp16x3', 3e-6), ('fp16', 2e-3), ('fp16f8', 1e-4)])
@pytest.mark.parametrize('B,C0,C1,Cout,H,W,norm,temb,res', [
    (2, 96, 0, 96, 32, 32, True, True, False),       # ResnetBlock Conv_0: GroupNorm + SiLU prologue, + Dense(temb)
    (3, 96, 96, 96, 16, 48, True, False, True),      # up-path block: virtual concat of two sources, residual, odd batch
    (1, 64, 32, 192, 32, 16, True, True, True),      # two cout groups, unequal sources
    (2, 32, 0, 96, 16, 16, False, False, False),     # no GroupNorm: the convolution reads x as it is
    (2, 128, 0, 128, 32, 32, True, True, False),     # 64-cout groups (conv_xk NT = 2): nf = 128 networks
    (1, 128, 128, 256, 16, 32, True, False, True),   # ... four groups, 16 stages, concat + residual
    (2, 64, 0, 64, 16, 16, False, False, True),      # ... one group, four stages, raw operand + residual
    (1, 80, 0, 128, 16, 16, True, False, False),     # an odd number of stages: conv_ff keeps the layer
])
def test_conv3x3_block_fused_prologue(B, C0, C1, Cout, H, W, norm, temb, res, precision, tol):
    """csd_conv3x3_block (csrc/conv_ff.hip) = Conv3x3(SiLU(x*scale + shift)) + bias + temb + res, and its per-tile
    GroupNorm partials, against fp64 torch (reference models/layers.py:632-675)"""
    from conditional_score_diffusion_amd import ops
    if precision == 'fp16' and (C0 % 32 or C1 % 32):
        pytest.skip('the single-plane fp16 form stages 32 channels at a time')
    g = torch.Generator().manual_seed(B * 1000 + C0 + Cout + H)
    Cin = C0 + C1
    x = torch.randn(B, H, W, Cin, generator=g) * 2.0 + 0.3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn(Cout, generator=g)
    sc = torch.rand(B, Cin, generator=g) + 0.5 if norm else None
    sh = torch.randn(B, Cin, generator=g) * 0.5 if norm else None
    tv = torch.randn(B, Cout + 32, generator=g) if temb else None
    rv = torch.randn(B, H, W, Cout, generator=g) * 3.0 if res else None
    d = dev()
    x0 = x[..., :C0].contiguous().to(d)
    x1 = x[..., C0:].contiguous().to(d) if C1 else None
    tgpu = tv.to(d)[:, 16:] if temb else None           # (a column window of a wider table, like dense_all in the network)
    opt = lambda t: None if t is None else t.to(d)      # noqa: E731
    if temb:
        # the wrapper makes its inputs contiguous, which would drop the row stride: call the C ABI through a strided view
        from conditional_score_diffusion_amd import _lib
        from conditional_score_diffusion_amd._lib import check, current_stream, lib, ptr
        y = torch.empty(B, H, W, Cout, device=d)
        stats = torch.empty(B * (H // 16) * (W // 16), Cout, 2, dtype=torch.float64, device=d)
        scr = torch.empty(lib().csd_conv3x3_block_scratch_bytes(Cin, Cout), dtype=torch.uint8, device=d)
        tfull = tv.to(d)
        tptr = tfull.data_ptr() + 16 * 4
        import ctypes
        wd, bd, scd, shd, rd = w.to(d), bias.to(d), opt(sc), opt(sh), opt(rv)      # (keep the device copies alive over the call)
        check(lib().csd_conv3x3_block(ptr(x0), ptr(x1), ptr(wd), ptr(bd), ptr(scd), ptr(shd),
                                      ctypes.c_void_p(tptr), tfull.shape[1], ptr(rd), 0.5, ptr(y), ptr(stats), B, C0, C1, Cout,
                                      H, W, _lib.PREC_IDS[precision], ptr(scr), current_stream(d)), 'conv3x3_block')
        torch.cuda.synchronize()
        tref = tv[:, 16:16 + Cout]
    else:
        y, stats = ops.conv3x3_block(x0, w.to(d), bias.to(d), x1=x1, nscale=opt(sc), nshift=opt(sh), res=opt(rv), out_scale=0.5,
                                     precision=precision, want_stats=True)
        tref = None
    xd = x.double()
    if norm:
        xd = torch.nn.functional.silu(xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :])
    ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if tref is not None:
        ref = ref + tref.double()[:, None, None, :]
    if rv is not None:
        ref = ref + rv.double()
    ref = ref * 0.5
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    # per-tile partials: (sum, sum of squares) of the tile's 256 pixels per cout, tiles in (sample, tile row, tile col) order
    yt = y.cpu().double().reshape(B, H // 16, 16, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(-1, 256, Cout)
    st = stats.cpu()
    assert (st[:, :, 0] - yt.sum(1)).abs().max().item() < 1e-4 * yt.abs().sum(1).max().item()
    assert (st[:, :, 1] - (yt * yt).sum(1)).abs().max().item() < 1e-5 * (yt * yt).sum(1).max().item()
