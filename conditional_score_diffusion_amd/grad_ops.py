"""Differentiable layer operators: ``torch.autograd.Function`` shells whose forward AND backward are HIP kernels
behind the C ABI (include/csd.h, csrc/backward.hip).

This is the training side of the path: the reference obtains these backward passes from
torch autograd over ATen; here autograd only records the graph - every gradient is computed by libcsd_hip.so:

=====================  ==========================================================================================
conv2d (3x3 / 1x1)     dX = csd_conv2d(dY, flip(W)^T) (stride 2: dY zero-inserted by csd_upfirdn2d; nearest-x2: 2x2
                       sum of the result by csd_upfirdn2d);  dW = csd_conv2d_wgrad;  db = csd_sum_inner + csd_sum_rows
groupnorm_act          csd_groupnorm_act_backward (+ csd_sum_rows for dgamma / dbeta)
attention              csd_attention_backward
linear                 csd_bgemm (dIn, dW) + csd_act (activation derivative) + csd_sum_rows (db)
bias_add_nchw          dX = dY;  dbias[b, c] = csd_sum_inner
axpby / scale_rows     themselves
dropout                csd_dropout (Philox mask) / csd_mul
sumsq_rows             per-sample squared L2 norm (the score-matching residual), dX = 2 g_b x
=====================  ==========================================================================================

torch moves data only (weight flips / transposes are copies, ``cat`` / slicing are views or copies).  There is no
CPU fallback: every function requires float32 GPU tensors.
"""
import torch

from . import _lib, ops
from ._lib import check, current_stream, lib, ptr


def _sum_inner(x, rows):
    x = x.contiguous()
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().csd_sum_inner(ptr(x), ptr(out), rows, x.numel() // rows, current_stream(x.device)), 'sum_inner')
    return out


def _sum_rows(x):
    x = x.contiguous()
    R, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    check(lib().csd_sum_rows(ptr(x), ptr(out), R, C, current_stream(x.device)), 'sum_rows')
    return out


def _act(x, act, dy=None):
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().csd_act(ptr(x), ptr(dy.contiguous()) if dy is not None else None, ptr(out), _lib.ACT_IDS[act], x.numel(),
                        current_stream(x.device)), 'act')
    return out


def _mul(a, b):
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib().csd_mul(ptr(a), ptr(b), ptr(out), a.numel(), current_stream(a.device)), 'mul')
    return out


def bgemm(A, B, out, M, N, K, sa, sb, sc, batch=1, z=(0, 0, 0), alpha=1.0):
    """out[z][m*sc0 + n*sc1] = alpha * sum_k A[z][m*sa0 + k*sa1] * B[z][k*sb0 + n*sb1] on raw storage."""
    check(lib().csd_bgemm(ptr(A), ptr(B), ptr(out), M, N, K, sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], batch, z[0], z[1], z[2],
                          float(alpha), current_stream(out.device)), 'bgemm')
    return out


# ---------------------------------------------------------------------------------------------------------------
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, downsample_pad, up2, precision):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, downsample_pad, up2, precision, bias is not None)
        return ops.conv2d(x, weight, bias, stride=stride, downsample_pad=downsample_pad, up2=up2, precision=precision)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, dpad, up2, precision, has_bias = ctx.cfg
        dy = dy.contiguous()
        B, Cin, H, W = x.shape
        Cout, _, k, _ = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = weight.flip(2, 3).transpose(0, 1).contiguous()          # [Cin, Cout, k, k]: data movement only
            if stride == 2:
                # Downsample (pad (0,1,0,1), stride 2): dy lands on the odd positions of an H x W grid, then a pad-1 conv
                one = torch.ones(1, 1, dtype=torch.float32, device=dy.device)
                z = ops._upfirdn2d_raw(dy, one, (2, 2), (1, 1), (1, -1, 1, -1))
                dx = ops.conv2d(z, wt, None, precision=precision)
            else:
                dx = ops.conv2d(dy, wt, None, precision=precision)
                if up2:      # nearest x2 in front of the conv: every source pixel fed a 2x2 block
                    ones = torch.ones(2, 2, dtype=torch.float32, device=dy.device)
                    dx = ops._upfirdn2d_raw(dx, ones, (1, 1), (2, 2), (0, 0, 0, 0))
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            sc = ops._scratch(lib().csd_conv_wgrad_scratch_bytes(B, Cin, Cout, H, W, k, stride, int(up2)), x.device)
            check(lib().csd_conv2d_wgrad_ex(ptr(x), ptr(dy), ptr(dw), B, Cin, Cout, H, W, k, stride, 1 if dpad else 0, int(up2),
                                            0 if precision == 'fp32' else 4, ptr(sc), current_stream(x.device)), 'conv2d_wgrad')
        if has_bias and ctx.needs_input_grad[2]:
            db = _sum_rows(_sum_inner(dy, B * Cout).view(B, Cout))
        return dx, dw, db, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, downsample_pad=False, up2=False, precision='fp32'):
    return _Conv2d.apply(x.contiguous(), weight.contiguous(), bias, stride, downsample_pad, up2, precision)


def nin(x, W, b, precision='fp32'):
    """NIN (models/layers.py:555-564): 1x1 contraction with W [in, out]."""
    return conv2d(x, W.t().reshape(W.shape[1], W.shape[0], 1, 1), b, precision=precision)


# ---------------------------------------------------------------------------------------------------------------
class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (groups, eps, act)
        return ops.groupnorm_act(x, gamma, beta, groups=groups, eps=eps, act=act)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        groups, eps, act = ctx.cfg
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty(B, C, dtype=torch.float32, device=x.device)
        db = torch.empty(B, C, dtype=torch.float32, device=x.device)
        check(lib().csd_groupnorm_act_backward(ptr(x), ptr(gamma), ptr(beta), ptr(dy), ptr(dx), ptr(dg), ptr(db), B, C, H, W,
                                               groups, eps, _lib.ACT_IDS[act], current_stream(x.device)), 'groupnorm_act_backward')
        return dx, _sum_rows(dg), _sum_rows(db), None, None, None


def groupnorm_act(x, gamma, beta, groups=32, eps=1e-6, act='none'):
    return _GroupNormAct.apply(x.contiguous(), gamma.contiguous(), beta.contiguous(), groups, eps, act)


# ---------------------------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        ctx.save_for_backward(q, k, v)
        return ops.attention(q, k, v)

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        do = do.contiguous()
        B, C, H, W = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sc = ops._scratch(lib().csd_attention_backward_scratch_bytes(B, C, H, W), q.device)
        check(lib().csd_attention_backward(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), B, C, H, W, ptr(sc),
                                           current_stream(q.device)), 'attention_backward')
        return dq, dk, dv


def attention(q, k, v):
    return _Attention.apply(q.contiguous(), k.contiguous(), v.contiguous())


# ---------------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act_in):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (act_in, bias is not None)
        return ops.linear(x, weight, bias, act_in=act_in)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        act_in, has_bias = ctx.cfg
        dy = dy.contiguous()
        B, K = x.shape
        N = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(x)                                  # d act_in(x)[b,k] = sum_n dy[b,n] W[n,k]
            bgemm(dy, weight, da, B, K, N, (N, 1), (K, 1), (K, 1))
            dx = da if act_in == 'none' else _act(x, act_in, da)
        if ctx.needs_input_grad[1]:
            a = x if act_in == 'none' else _act(x, act_in)
            dw = torch.empty_like(weight)                             # dW[n,k] = sum_b dy[b,n] a[b,k]
            bgemm(dy, a, dw, N, K, B, (1, N), (K, 1), (K, 1))
        if has_bias and ctx.needs_input_grad[2]:
            db = _sum_rows(dy)
        return dx, dw, db, None


def linear(x, weight, bias=None, act_in='none'):
    return _Linear.apply(x.contiguous(), weight.contiguous(), bias, act_in)


# ---------------------------------------------------------------------------------------------------------------
class _BiasAdd(torch.autograd.Function):
    """x + bias[:, :, None, None], bias [B, C] (the time-embedding add of the residual blocks)."""

    @staticmethod
    def forward(ctx, x, bias):
        return ops.bias_add_nchw(x, bias)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, C = dy.shape[0], dy.shape[1]
        db = _sum_inner(dy, B * C).view(B, C) if ctx.needs_input_grad[1] else None
        return dy, db


def bias_add_nchw(x, bias):
    return _BiasAdd.apply(x.contiguous(), bias.contiguous())


class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta, gamma, post):
        ctx.cfg = (alpha, beta, post, b is not None)
        return ops.axpby(a, b, alpha, beta, gamma, post)

    @staticmethod
    def backward(ctx, dy):
        alpha, beta, post, has_b = ctx.cfg
        dy = dy.contiguous()
        da = dy if alpha * post == 1.0 else ops.axpby(dy, None, alpha * post, 0.0, 0.0, 1.0)
        db = None
        if has_b and ctx.needs_input_grad[1]:
            db = dy if beta * post == 1.0 else ops.axpby(dy, None, beta * post, 0.0, 0.0, 1.0)
        return da, db, None, None, None, None


def axpby(a, b=None, alpha=1.0, beta=1.0, gamma=0.0, post=1.0):
    return _Axpby.apply(a.contiguous(), None if b is None else b.contiguous(), alpha, beta, gamma, post)


class _ScaleRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, divide):
        ctx.save_for_backward(scale)
        ctx.divide = divide
        return ops.scale_rows(x, scale, divide=divide)

    @staticmethod
    def backward(ctx, dy):
        scale, = ctx.saved_tensors
        return ops.scale_rows(dy.contiguous(), scale, divide=ctx.divide), None, None


def scale_rows(x, scale, divide=False):
    """x[b] * scale[b] (or / scale[b]); differentiable in x (the scales are SDE constants)."""
    return _ScaleRows.apply(x.contiguous(), scale.contiguous(), divide)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, stream_id):
        out, mask = torch.empty_like(x), torch.empty_like(x)
        check(lib().csd_dropout(ptr(x), ptr(out), ptr(mask), float(p), int(seed), int(stream_id), x.numel(),
                                current_stream(x.device)), 'dropout')
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        return _mul(dy, mask), None, None, None


def dropout(x, p, seed, stream_id):
    """nn.Dropout(p) in training mode; the mask is a pure function of (seed, stream_id)."""
    if p <= 0.0:
        return x
    return _Dropout.apply(x.contiguous(), p, seed, stream_id)


class _SumSqRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        n = ops.row_norms(x)
        return _mul(n, n)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ops.scale_rows(x, ops.axpby(g.contiguous().float(), None, 2.0, 0.0, 0.0, 1.0))


def sumsq_rows(x):
    """[B] per-sample sum of squares (fp64 accumulation on the device), differentiable."""
    return _SumSqRows.apply(x.contiguous())
