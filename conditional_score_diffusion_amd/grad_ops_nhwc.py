"""The differentiable layer operators of grad_ops on NHWC activations ([B, H, W, C] fp32, contiguous).

The training graph of the DDPM family keeps its activations in the library's own layout, so a layer is exactly its kernels:
no NCHW<->NHWC change on either side of a convolution (12-16 % of the operator-granular NCHW step).  Same kernels, same
C ABI family (include/csd.h: csd_conv2d_ex / csd_conv2d_wgrad_ex layout flags, csd_*_nhwc); the network input and output keep
the reference's NCHW (``layout`` arguments of conv2d).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check, current_stream, lib, ptr
from .grad_ops import _sum_rows, axpby, dropout, linear  # noqa: F401  (layout-free operators are shared)

IN_NHWC, OUT_NHWC, SPLIT_BF16 = 1, 2, 4
WEIGHT_T = 4            # csd_conv2d_ex: the weight is the transposed convolution's (data gradient)


def _conv_raw(x, weight, bias, stride, dpad, up2, precision, layout):
    """csd_conv2d_ex; x is [B,H,W,Cin] if layout & 1 else [B,Cin,H,W]; the result [B,OH,OW,Cout] if layout & 2 else NCHW;
    layout & 4: ``weight`` is [Cin, Cout, k, k] (the forward weight of the convolution whose data gradient this is)."""
    if layout & IN_NHWC:
        B, H, W, Cin = x.shape
    else:
        B, Cin, H, W = x.shape
    if layout & WEIGHT_T:
        cin_w, Cout, k, _ = weight.shape
    else:
        Cout, cin_w, k, _ = weight.shape
    if cin_w != Cin:
        raise RuntimeError('conv2d: weight %s does not match %d input channels' % (tuple(weight.shape), Cin))
    s = 2 if up2 else 1
    OH, OW = H * s // stride, W * s // stride
    y = torch.empty((B, OH, OW, Cout) if layout & OUT_NHWC else (B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    sc = ops._scratch(lib().csd_conv_scratch_bytes(B, Cin, Cout, H, W, k, int(up2)), x.device)
    check(lib().csd_conv2d_ex(ptr(x), ptr(weight), ptr(bias), ptr(y), B, Cin, Cout, H, W, k, stride, 1 if dpad else 0, int(up2),
                              _lib.PREC_IDS[precision], layout, ptr(sc), current_stream(x.device)), 'conv2d_ex')
    return y


def _sum_pixels(x):
    """[B, H, W, C] -> [B, C]"""
    B, H, W, C = x.shape
    out = torch.empty(B, C, dtype=torch.float32, device=x.device)
    sc = ops._scratch(lib().csd_sum_pixels_scratch_bytes(B, H * W, C), x.device)
    check(lib().csd_sum_pixels_nhwc(ptr(x), ptr(out), B, H * W, C, ptr(sc), current_stream(x.device)), 'sum_pixels_nhwc')
    return out


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, dpad, up2, precision, layout):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, dpad, up2, precision, bias is not None, layout)
        return _conv_raw(x, weight, bias, stride, dpad, up2, precision, layout)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, dpad, up2, precision, has_bias, layout = ctx.cfg
        dy = dy.contiguous()
        in_nhwc, out_nhwc = bool(layout & IN_NHWC), bool(layout & OUT_NHWC)
        if in_nhwc:
            B, H, W, Cin = x.shape
        else:
            B, Cin, H, W = x.shape
        Cout, _, k, _ = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if not in_nhwc:
                raise RuntimeError('the data gradient of an NCHW-input convolution is not needed by the training graph')
            wt = weight                       # used transposed + flipped by the pack kernel (WEIGHT_T): nothing is materialised
            lay = (IN_NHWC if out_nhwc else 0) | OUT_NHWC | WEIGHT_T
            if stride == 2:
                if not out_nhwc:
                    raise RuntimeError('stride-2 convolutions live inside the NHWC graph')
                z = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dy.device)
                check(lib().csd_zero_insert_odd_nhwc(ptr(dy), ptr(z), B, H // 2, W // 2, Cout, current_stream(dy.device)),
                      'zero_insert_odd_nhwc')
                dx = _conv_raw(z, wt, None, 1, False, False, precision, lay)
            else:
                dx = _conv_raw(dy, wt, None, 1, False, False, precision, lay)
                if up2:
                    full = dx
                    dx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device)
                    check(lib().csd_sumpool2_nhwc(ptr(full), ptr(dx), B, H, W, Cin, current_stream(dy.device)), 'sumpool2_nhwc')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            sc = ops._scratch(lib().csd_conv_wgrad_scratch_bytes(B, Cin, Cout, H, W, k, stride, int(up2)), x.device)
            check(lib().csd_conv2d_wgrad_ex(ptr(x), ptr(dy), ptr(dw), B, Cin, Cout, H, W, k, stride, 1 if dpad else 0, int(up2),
                                            layout | (0 if precision == 'fp32' else SPLIT_BF16), ptr(sc),
                                            current_stream(x.device)), 'conv2d_wgrad_ex')
        if has_bias and ctx.needs_input_grad[2]:
            if out_nhwc and Cout % 4 == 0:
                db = _sum_rows(_sum_pixels(dy))
            else:
                from .grad_ops import _sum_inner
                rows = dy.permute(0, 3, 1, 2).contiguous() if out_nhwc else dy
                db = _sum_rows(_sum_inner(rows, B * Cout).view(B, Cout))
        return dx, dw, db, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, downsample_pad=False, up2=False, precision='fp32', layout=IN_NHWC | OUT_NHWC):
    return _Conv2d.apply(x.contiguous(), weight.contiguous(), bias, stride, downsample_pad, up2, precision, layout)


def nin(x, W, b, precision='fp32'):
    return conv2d(x, W.t().reshape(W.shape[1], W.shape[0], 1, 1), b, precision=precision)


class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        rs = torch.empty(B, C, dtype=torch.float32, device=x.device)
        ms = torch.empty(B, C, dtype=torch.float32, device=x.device)
        sc = ops._scratch(lib().csd_groupnorm_nhwc_scratch_bytes(B, C, H * W), x.device)
        check(lib().csd_groupnorm_act_nhwc(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(rs), ptr(ms), B, C, H * W, groups, eps,
                                           _lib.ACT_IDS[act], ptr(sc), current_stream(x.device)), 'groupnorm_act_nhwc')
        ctx.save_for_backward(x, gamma, beta, rs, ms)
        ctx.cfg = (groups, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, rs, ms = ctx.saved_tensors
        groups, act = ctx.cfg
        dy = dy.contiguous()
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        rows = torch.empty(B, 2 * C, dtype=torch.float32, device=x.device)       # [dgamma row | dbeta row] per sample: ONE batch reduction
        sc = ops._scratch(lib().csd_groupnorm_nhwc_scratch_bytes(B, C, H * W), x.device)
        check(lib().csd_groupnorm_act_backward_nhwc(ptr(x), ptr(gamma), ptr(beta), ptr(rs), ptr(ms), ptr(dy), ptr(dx), ptr(rows),
                                                    ctypes.c_void_p(rows.data_ptr() + 4 * C), 2 * C, B, C, H * W, groups,
                                                    _lib.ACT_IDS[act], ptr(sc), current_stream(x.device)),
              'groupnorm_act_backward_nhwc')
        both = _sum_rows(rows)
        return dx, both[:C], both[C:], None, None, None


def groupnorm_act(x, gamma, beta, groups=32, eps=1e-6, act='none'):
    return _GroupNormAct.apply(x.contiguous(), gamma.contiguous(), beta.contiguous(), groups, eps, act)


class _Attention(torch.autograd.Function):
    """qkv [B, H, W, 3C] (q | k | v per pixel) -> [B, H, W, C]"""

    @staticmethod
    def forward(ctx, qkv):
        B, H, W, C3 = qkv.shape
        out = torch.empty(B, H, W, C3 // 3, dtype=torch.float32, device=qkv.device)
        check(lib().csd_attention_nhwc(ptr(qkv), ptr(out), B, H * W, C3 // 3, current_stream(qkv.device)), 'attention_nhwc')
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, = ctx.saved_tensors
        do = do.contiguous()
        B, H, W, C3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        sc = ops._scratch(lib().csd_attention_backward_scratch_bytes(B, C3 // 3, H * W, 1), qkv.device)
        check(lib().csd_attention_backward_nhwc(ptr(qkv), ptr(do), ptr(dqkv), B, H * W, C3 // 3, ptr(sc),
                                                current_stream(qkv.device)), 'attention_backward_nhwc')
        return dqkv


def attention(qkv):
    return _Attention.apply(qkv.contiguous())


class _BiasAdd(torch.autograd.Function):
    """x [B,H,W,C] + bias[b, c]"""

    @staticmethod
    def forward(ctx, x, bias):
        B, H, W, C = x.shape
        out = torch.empty_like(x)
        check(lib().csd_bias_add_nhwc(ptr(x), ptr(bias), ptr(out), B, H * W, C, current_stream(x.device)), 'bias_add_nhwc')
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return dy, (_sum_pixels(dy) if ctx.needs_input_grad[1] else None)


def bias_add(x, bias):
    return _BiasAdd.apply(x.contiguous(), bias.contiguous())
