"""Thin tensor-level wrappers over the per-operator C ABI (include/csd.h).

Tensors are containers only: every function checks that its operands are contiguous fp32 GPU
tensors, allocates the output / scratch with torch, and enqueues the HIP kernels on the current
stream.  No torch arithmetic happens here.
"""
import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr, require_gpu_tensor


def _scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _c(t, name):
    require_gpu_tensor(t, name)
    return t.contiguous()


def groupnorm_act(x, gamma, beta, groups=32, eps=1e-6, act='none'):
    """act(GroupNorm(x)) - nn.GroupNorm + get_act (models/layers.py:571,638,646)."""
    x, gamma, beta = _c(x, 'x'), _c(gamma, 'gamma'), _c(beta, 'beta')
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    sc = _scratch(lib().csd_groupnorm_scratch_bytes(B, C, H, W), x.device)
    check(lib().csd_groupnorm_act(ptr(x), ptr(gamma), ptr(beta), ptr(y), B, C, H, W, groups, eps,
                                  _lib.ACT_IDS[act], ptr(sc), current_stream(x.device)), 'groupnorm_act')
    return y


def conv2d(x, weight, bias=None, stride=1, downsample_pad=False, up2=False, precision='fp32'):
    """3x3 / 1x1 convolution, weight OIHW (models/layers.py:100-132); ``stride=2,
    downsample_pad=True`` is the reference Downsample (pad (0,1,0,1), models/layers.py:619-625);
    ``up2`` applies nearest x2 first (models/layers.py:600-604)."""
    x, weight = _c(x, 'x'), _c(weight, 'weight')
    if bias is not None:
        bias = _c(bias, 'bias')
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    if Cin_w != Cin or kh != kw:
        raise RuntimeError('conv2d: weight %s does not match input channels %d' % (tuple(weight.shape), Cin))
    OH = (H * (2 if up2 else 1)) // stride
    OW = (W * (2 if up2 else 1)) // stride
    y = torch.empty(B, Cout, OH, OW, dtype=torch.float32, device=x.device)
    sc = _scratch(lib().csd_conv_scratch_bytes(B, Cin, Cout, H, W, kh, int(up2)), x.device)
    check(lib().csd_conv2d(ptr(x), ptr(weight), ptr(bias), ptr(y), B, Cin, Cout, H, W, kh, stride,
                           1 if downsample_pad else 0, int(up2), _lib.PREC_IDS[precision], ptr(sc),
                           current_stream(x.device)), 'conv2d')
    return y


def conv3x3_block(x0, weight, bias=None, x1=None, nscale=None, nshift=None, temb=None, res=None, out_scale=1.0,
                  precision='fp16x3', want_stats=False):
    """The ResnetBlock convolution with its prologue fused (models/layers.py:632-675): y = Conv3x3(SiLU(x*nscale + nshift))
    (+ bias + temb[:, None, None, :] + res) * out_scale on NHWC fp32 tensors; x = x0 (| x1: virtual concat).  csd_conv3x3_block.
    Returns y [B,H,W,Cout] (and the per-tile (sum, sum of squares) partials of y when ``want_stats``)."""
    x0, weight = _c(x0, 'x0'), _c(weight, 'weight')
    B, H, W, C0 = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = _c(x1, 'x1')
        C1 = x1.shape[3]
    Cout = weight.shape[0]
    if tuple(weight.shape) != (Cout, C0 + C1, 3, 3):
        raise RuntimeError('conv3x3_block: weight %s does not match %d input channels' % (tuple(weight.shape), C0 + C1))
    opt = [None if t is None else _c(t, 't') for t in (bias, nscale, nshift, temb, res)]
    bias, nscale, nshift, temb, res = opt
    y = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x0.device)
    stats = torch.empty(B * ((H + 15) // 16) * ((W + 15) // 16), Cout, 2, dtype=torch.float64, device=x0.device) if want_stats else None
    sc = _scratch(lib().csd_conv3x3_block_scratch_bytes(C0 + C1, Cout), x0.device)
    check(lib().csd_conv3x3_block(ptr(x0), ptr(x1), ptr(weight), ptr(bias), ptr(nscale), ptr(nshift), ptr(temb),
                                  temb.shape[1] if temb is not None else 0, ptr(res), float(out_scale), ptr(y), ptr(stats),
                                  B, C0, C1, Cout, H, W, _lib.PREC_IDS[precision], ptr(sc), current_stream(x0.device)),
          'conv3x3_block')
    return (y, stats) if want_stats else y


def attention(q, k, v):
    """softmax(q.k C^-1/2) v over H*W positions (models/layers.py:584-588)."""
    q, k, v = _c(q, 'q'), _c(k, 'k'), _c(v, 'v')
    B, C, H, W = q.shape
    out = torch.empty_like(q)
    sc = _scratch(lib().csd_attention_scratch_bytes(B, C, H, W), q.device)
    check(lib().csd_attention(ptr(q), ptr(k), ptr(v), ptr(out), B, C, H, W, ptr(sc), current_stream(q.device)),
          'attention')
    return out


def _upfirdn2d_raw(x, kernel, up, down, pad4):
    """csd_upfirdn2d with per-axis factors and four pads (x0, x1, y0, y1); negative pads crop."""
    x, kernel = _c(x, 'x'), _c(kernel, 'kernel')
    N, C, H, W = x.shape
    kh, kw = kernel.shape
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad4
    OH = (H * uy + py0 + py1 - kh) // dy + 1
    OW = (W * ux + px0 + px1 - kw) // dx + 1
    out = torch.empty(N, C, OH, OW, dtype=torch.float32, device=x.device)
    check(lib().csd_upfirdn2d(ptr(x), ptr(kernel), ptr(out), N, C, H, W, kh, kw, ux, uy, dx, dy, px0, px1, py0, py1,
                              current_stream(x.device)), 'upfirdn2d')
    return out


class _UpFirDn2dBackward(torch.autograd.Function):
    """grad_input = upfirdn2d(grad_output, flip(kernel), up=down, down=up, pad=g_pad) (op/upfirdn2d.py:20-85)."""

    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad, ctx.in_size, ctx.out_size = up, down, pad, in_size, out_size
        g = _upfirdn2d_raw(grad_output.reshape(in_size[0], in_size[1], out_size[0], out_size[1]), grad_kernel, down, up, g_pad)
        return g.view(in_size)

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gg = _upfirdn2d_raw(gradgrad_input.contiguous(), kernel, ctx.up, ctx.down, ctx.pad)
        return gg, None, None, None, None, None, None, None, None


class _UpFirDn2d(torch.autograd.Function):
    """Differentiable upfirdn2d, same autograd structure as the reference's UpFirDn2d (op/upfirdn2d.py:88-145)."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        ux, uy = up
        dx, dy = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, _, in_h, in_w = x.shape
        out = _upfirdn2d_raw(x, kernel, up, down, pad)
        out_h, out_w = out.shape[2], out.shape[3]
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        ctx.in_size, ctx.out_size, ctx.up, ctx.down, ctx.pad = tuple(x.shape), (out_h, out_w), up, down, pad
        ctx.g_pad = (kw - px0 - 1, in_w * ux - out_w * dx + px0 - ux + 1,
                     kh - py0 - 1, in_h * uy - out_h * dy + py0 - uy + 1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        g = _UpFirDn2dBackward.apply(grad_output.contiguous(), kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad,
                                     ctx.in_size, ctx.out_size)
        return g, None, None, None, None


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Same call signature as the reference's op.upfirdn2d (op/upfirdn2d.py:147-158); differentiable w.r.t. ``x`` with
    the reference's backward (the same FIR kernel flipped, up/down swapped, pads g_pad)."""
    return _UpFirDn2d.apply(x, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """Forward of op.fused_leaky_relu (op/fused_act.py:86-97): lrelu(x + b[c]) * scale."""
    x, bias = _c(x, 'x'), _c(bias, 'bias')
    out = torch.empty_like(x)
    inner = 1
    for s in x.shape[2:]:
        inner *= s
    check(lib().csd_fused_bias_act(ptr(x), ptr(bias), None, ptr(out), x.numel(), x.shape[1], inner, 3, 0,
                                   negative_slope, scale, current_stream(x.device)), 'fused_bias_act')
    return out


def nearest_up2(x):
    x = _c(x, 'x')
    N, C, H, W = x.shape
    out = torch.empty(N, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    check(lib().csd_nearest_up2(ptr(x), ptr(out), N, C, H, W, current_stream(x.device)), 'nearest_up2')
    return out


def timestep_embedding(t, dim):
    t = _c(t, 't')
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    check(lib().csd_timestep_embedding(ptr(t), ptr(out), t.shape[0], dim, current_stream(t.device)),
          'timestep_embedding')
    return out


def linear(x, weight, bias=None, act_in='none'):
    """act_in(x) @ weight.T + bias on [B, K] (nn.Linear; the reference applies the activation to the INPUT of the
    second temb layer and of every Dense_0: models/ncsnpp.py:257-263, layerspp.py:253-255)."""
    x, weight = _c(x, 'x'), _c(weight, 'weight')
    if bias is not None:
        bias = _c(bias, 'bias')
    B, K = x.shape
    N = weight.shape[0]
    out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(lib().csd_linear(ptr(x), ptr(weight), ptr(bias), ptr(out), B, K, N, _lib.ACT_IDS[act_in],
                           current_stream(x.device)), 'linear')
    return out


def fourier_embedding(t, W):
    """GaussianFourierProjection (models/layerspp.py:32-41): [sin(2 pi W t), cos(2 pi W t)]."""
    t, W = _c(t, 't'), _c(W, 'W')
    out = torch.empty(t.shape[0], 2 * W.shape[0], dtype=torch.float32, device=t.device)
    check(lib().csd_fourier_embedding(ptr(t), ptr(W), ptr(out), t.shape[0], W.shape[0], current_stream(t.device)),
          'fourier_embedding')
    return out


def axpby(a, b=None, alpha=1.0, beta=1.0, gamma=0.0, post=1.0):
    """(alpha*a + beta*b + gamma) * post, elementwise (b optional)."""
    a = _c(a, 'a')
    if b is not None:
        b = _c(b, 'b')
        if b.shape != a.shape:
            raise RuntimeError('axpby: shapes %s and %s differ' % (tuple(a.shape), tuple(b.shape)))
    out = torch.empty_like(a)
    check(lib().csd_axpby(ptr(a), ptr(b), ptr(out), float(alpha), float(beta), float(gamma), float(post), a.numel(),
                          current_stream(a.device)), 'axpby')
    return out


def bias_add_nchw(x, bias, act='none'):
    """act(x + bias[:, :, None, None]) with bias [B, C] (or [C])."""
    x, bias = _c(x, 'x'), _c(bias, 'bias')
    B, C = x.shape[0], x.shape[1]
    inner = x.numel() // (B * C)
    stride = C if bias.dim() == 2 else 0
    out = torch.empty_like(x)
    check(lib().csd_bias_add_nchw(ptr(x), ptr(bias), ptr(out), B, C, inner, stride, _lib.ACT_IDS[act],
                                  current_stream(x.device)), 'bias_add_nchw')
    return out


def randn(shape, seed, stream_id, device):
    """Counter-based standard normals (Philox4x32-10): same (seed, stream_id) -> same tensor."""
    out = torch.empty(*shape, dtype=torch.float32, device=device)
    check(lib().csd_randn(ptr(out), out.numel(), int(seed), int(stream_id), current_stream(out.device)), 'randn')
    return out


def scale_rows(x, scale, divide=False):
    """x[b] * scale[b] (or / scale[b]) - divide_by_sigmas (models/utils.py:50-74)."""
    x, scale = _c(x, 'x'), _c(scale, 'scale')
    out = torch.empty_like(x)
    B = x.shape[0]
    check(lib().csd_scale_rows(ptr(out), ptr(x), ptr(scale), int(divide), B, x.numel() // B,
                               current_stream(x.device)), 'scale_rows')
    return out


def langevin_step(x, net, z, std, snr, alpha=1.0):
    """In-place Langevin corrector update (sampling/correctors.py:51-78,88-108); ``alpha`` = sde.alphas[timestep] for the VP / subVP
    SDEs, 1 for the VE SDEs; returns (x, x_mean)."""
    x, net, z = _c(x, 'x'), _c(net, 'net'), _c(z, 'z')
    B = x.shape[0]
    x_mean = torch.empty_like(x)
    sc = _scratch(lib().csd_update_scratch_bytes(B), x.device)
    check(lib().csd_langevin_step(ptr(x), ptr(x_mean), ptr(net), ptr(z), float(std), float(snr), float(alpha), B,
                                  x.numel() // B, ptr(sc), current_stream(x.device)), 'langevin_step')
    return x, x_mean


def row_norms(x):
    """||x_b||_2 per sample -> [B] (fp64 accumulation on the device)."""
    x = _c(x, 'x')
    B = x.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib().csd_row_norms(ptr(x), ptr(out), B, x.numel() // B, current_stream(x.device)), 'row_norms')
    return out


def affine_noise_step(x, score, z, p, a, c):
    """In-place x_mean = p*x + a*score; x = x_mean + c*z (Euler-Maruyama / ancestral / annealed-Langevin updates)."""
    x, score, z = _c(x, 'x'), _c(score, 'score'), _c(z, 'z')
    x_mean = torch.empty_like(x)
    check(lib().csd_affine_noise_step(ptr(x), ptr(x_mean), ptr(score), ptr(z), float(p), float(a), float(c),
                                      x.numel(), current_stream(x.device)), 'affine_noise_step')
    return x, x_mean


def reverse_diffusion_step(x, net, z, std, G):
    """In-place reverse-diffusion predictor update (sampling/predictors.py:97-102)."""
    x, net, z = _c(x, 'x'), _c(net, 'net'), _c(z, 'z')
    B = x.shape[0]
    x_mean = torch.empty_like(x)
    check(lib().csd_reverse_diffusion_step(ptr(x), ptr(x_mean), ptr(net), ptr(z), float(std), float(G), B,
                                           x.numel() // B, current_stream(x.device)), 'reverse_diffusion_step')
    return x, x_mean
