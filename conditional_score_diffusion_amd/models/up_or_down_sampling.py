"""FIR resampling functions of the reference (models/up_or_down_sampling.py:59-69,137-257) on the HIP upfirdn2d kernel: same names,
arguments and defaults; differentiable (ops.upfirdn2d carries the reference's backward / double backward).

``upsample_conv_2d`` (:72-134) is not provided: it fails upstream and no configuration reaches it.
"""
import numpy as np
import torch

from .. import grad_ops, ops


def _setup_kernel(k):
    """:181-189 - separable taps -> normalised square 2-D kernel (numpy float32)."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    assert k.ndim == 2
    assert k.shape[0] == k.shape[1]
    return k


def naive_upsample_2d(x, factor=2):
    """:59-63 - nearest neighbour = zero insertion followed by a factor x factor box filter."""
    box = torch.ones(factor, factor, dtype=torch.float32, device=x.device)
    return ops.upfirdn2d(x, box, up=factor, pad=(factor - 1, 0))


def naive_downsample_2d(x, factor=2):
    """:66-69 - mean over factor x factor blocks."""
    box = torch.full((factor, factor), 1.0 / (factor * factor), dtype=torch.float32, device=x.device)
    return ops.upfirdn2d(x, box, down=factor, pad=(0, 0))


def upsample_2d(x, k=None, factor=2, gain=1):
    """:196-226."""
    assert isinstance(factor, int) and factor >= 1
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * (gain * (factor ** 2))
    p = k.shape[0] - factor
    return ops.upfirdn2d(x, torch.tensor(k, device=x.device), up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k=None, factor=2, gain=1):
    """:229-257."""
    assert isinstance(factor, int) and factor >= 1
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * gain
    p = k.shape[0] - factor
    return ops.upfirdn2d(x, torch.tensor(k, device=x.device), down=factor, pad=((p + 1) // 2, p // 2))


def conv_downsample_2d(x, w, k=None, factor=2, gain=1, precision='fp32'):
    """:137-178 - FIR filter, then a VALID stride-``factor`` convolution with ``w`` [out, in, 3, 3] (factor 2, the form
    layerspp.Downsample(with_conv=True, fir=True) uses): run as the library's stride-2 convolution (pad (0,1,0,1)) on a FIR output
    one pixel wider than the reference's - its first H/2 rows / columns read exactly the valid windows - and cropped."""
    assert isinstance(factor, int) and factor >= 1
    _outC, _inC, convH, convW = w.shape
    assert convW == convH
    if factor != 2 or convW != 3:
        raise NotImplementedError('conv_downsample_2d on the HIP path: factor 2 with a 3x3 weight')
    if k is None:
        k = [1] * factor
    k = _setup_kernel(k) * gain
    p = (k.shape[0] - factor) + (convW - 1)
    z = ops.upfirdn2d(x, torch.tensor(k, device=x.device), pad=((p + 1) // 2, p // 2 + 1))
    y = grad_ops.conv2d(z, w, None, stride=2, downsample_pad=True, precision=precision)
    return y[:, :, :-1, :-1].contiguous()
