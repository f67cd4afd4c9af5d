"""op/fused_act.py of the reference (:20-97) on csd_fused_bias_act: ``fused_leaky_relu`` / ``FusedLeakyReLU`` with the same autograd
structure - forward ``lrelu(x + b, slope) * scale``; backward w.r.t. x from the saved OUTPUT (grad mode 1 of the kernel), bias gradient
= sum over batch and trailing dims (HIP reductions); the backward itself is differentiable (double backward)."""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib
from .._lib import check, current_stream, lib, ptr
from ..grad_ops import _sum_inner, _sum_rows


def _fba(x, bias, ref, act, grad, alpha, scale):
    """csd_fused_bias_act (op/fused_bias_act_kernel.cu:18-49): x [N, C, ...]"""
    _lib.require_gpu_tensor(x, 'input')
    x = x.contiguous()
    out = torch.empty_like(x)
    C = x.shape[1]
    inner = x.numel() // (x.shape[0] * C)
    check(lib().csd_fused_bias_act(ptr(x), ptr(bias.contiguous()) if bias is not None else None,
                                   ptr(ref.contiguous()) if ref is not None else None, ptr(out), x.numel(), C, inner, act, grad,
                                   float(alpha), float(scale), current_stream(x.device)), 'fused_bias_act')
    return out


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = _fba(grad_output, None, out, 3, 1, negative_slope, scale)
        N, C = grad_input.shape[0], grad_input.shape[1]
        grad_bias = _sum_rows(_sum_inner(grad_input, N * C).view(N, C))          # sum over dim 0 and dims 2..
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gradgrad_out = _fba(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = _fba(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output.contiguous(), out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """(The reference's CPU branch hard-codes slope 0.2, op/fused_act.py:89-94; the kernel branch - mirrored here - uses the argument.)"""
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
