"""Numerics study (test infrastructure, CPU only): how accurate is a 3x3 convolution whose operands are split x = hi + lo
(hi = fp16(x)) when only hi*hi runs on the fp16 matrix cores and the two correction products hi_w*lo_x + lo_w*hi_x run with
8-bit (OCP e4m3) operands on the fp8 matrix cores?  Emulated here on the oracle's SR3-160 network (torch CPU, fp32 accumulate)
against the plain fp32 forward, next to the emulation of the modes that exist (fp16: hi*hi only; fp16x3: both corrections in fp16)
so the emulation can be calibrated on the errors measured on the MI355X (tests/test_gpu_fullsize.py).

    python oracle/fp8_correction_study.py [--size 160] [--which all|3x3s1]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
import score_oracle as so  # noqa: E402

SCALE = 2.0 ** 12          # lo operands are scaled into the e4m3 range before rounding (a power of two: exact)
WS = 2.0 ** 8              # the library pre-scales weights by 2^8 (keeps the lo plane out of the fp16 subnormals)


def q8(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


REAL_CONV2D = F.conv2d      # (score_oracle.F is torch.nn.functional itself: keep the genuine function)


def make_conv(mode, which):
    real = REAL_CONV2D

    def conv(x, w, b=None, stride=1, padding=0, **kw):
        k = w.shape[-1]
        studied = k == 3 and stride == 1 and w.shape[1] % 16 == 0 and w.shape[0] % 96 == 0
        if mode == 'fp32' or (which == '3x3s1' and not studied):
            # (layers outside the studied kernel run in the certified split mode, error ~1e-6: treated as exact)
            return real(x, w, b, stride=stride, padding=padding, **kw)
        xh = x.half().float()
        wh = (w * WS).half().float()
        y = real(xh, wh, None, stride=stride, padding=padding, **kw)
        if mode != 'fp16':
            xl, wl = x - xh, w * WS - wh
            if mode == 'fp16x3':
                y = y + real(xl.half().float(), wh, None, stride=stride, padding=padding, **kw) \
                      + real(xh, wl.half().float(), None, stride=stride, padding=padding, **kw)
            elif mode == 'fp16+fp8':
                c = real(q8(xl * SCALE), q8(wh), None, stride=stride, padding=padding, **kw) \
                    + real(q8(xh), q8(wl * SCALE), None, stride=stride, padding=padding, **kw)
                y = y + c / SCALE
            else:
                raise ValueError(mode)
        y = y / WS
        return y if b is None else y + b.reshape(1, -1, 1, 1)
    return conv


def errs(a, b):
    a, b = a.double(), b.double()
    rms = float(b.pow(2).mean().sqrt())
    return float((a - b).norm() / b.norm()), float(((a - b).abs() / (b.abs() + rms)).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=160)
    ap.add_argument('--which', default='all')
    a = ap.parse_args()
    torch.set_num_threads(16)
    kw = dict(cases.SR3_160)
    kw['image_size'] = a.size
    cfg = cases.make_config(**kw)
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    rs = np.random.RandomState(3)
    B = 1
    S = a.size
    x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32)) * 20.
    y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32))
    lab = torch.full((B,), 600.)
    outs = {}
    for mode in ('fp32', 'fp16', 'fp16x3', 'fp16+fp8'):
        so.F.conv2d = make_conv(mode, a.which)
        try:
            with torch.no_grad():
                outs[mode] = so.paired_forward(p, nc, x, y, lab, True)
        finally:
            so.F.conv2d = REAL_CONV2D
        if mode != 'fp32':
            n, e = errs(outs[mode], outs['fp32'])
            print('%-9s vs fp32: norm-wise %.3e  element-wise %.3e' % (mode, n, e), flush=True)


if __name__ == '__main__':
    main()
