"""Micro-probe of the fused-prologue block convolution (csd_conv3x3_block): the SR3-160 layer shapes at the bench batch, timed with
HIP events; also the target of rocprofv3 --kernel-trace / --pmc runs.     REPS=20 PREC=fp16x3 python tools/ff_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from conditional_score_diffusion_amd import ops  # noqa: E402

SHAPES = [  # B, C0, C1, Cout, H, residual
    (64, 96, 0, 96, 160, False),
    (64, 96, 0, 96, 160, True),
    (64, 96, 96, 96, 160, False),
    (64, 96, 0, 96, 80, True),
    (64, 192, 96, 96, 80, False),
    (64, 192, 0, 192, 80, True),
    (64, 128, 0, 128, 64, True),       # 6, 7: nf = 128 networks (64-cout groups)
    (64, 256, 128, 128, 64, False),
]


def main():
    reps = int(os.environ.get('REPS', '10'))
    only = os.environ.get('ONLY')
    dev = torch.device('cuda:0')
    for prec in os.environ.get('PREC', 'fp16x3,fp16').split(','):
        for i, (B, C0, C1, Cout, H, res) in enumerate(SHAPES):
            if only is not None and str(i) not in only.split(','):
                continue
            Cin = C0 + C1
            x0 = torch.randn(B, H, H, C0, device=dev)
            x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (Cin * 9)) ** 0.5
            b = torch.randn(Cout, device=dev)
            sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
            r = torch.randn(B, H, H, Cout, device=dev) if res else None
            y, st = ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision=prec, want_stats=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision=prec, want_stats=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            fl = 2.0 * B * H * H * Cout * Cin * 9
            by = B * H * H * (Cin * 4.0 + Cout * 4.0 * (2 if res else 1))
            print('%-7s C %3d+%-3d -> %3d @%3d res=%d : %7.1f us (incl. ~weight pack)  %6.1f TF/s alg  %5.2f TB/s alg' %
                  (prec, C0, C1, Cout, H, res, us, fl / us / 1e6, by / us / 1e6))


if __name__ == '__main__':
    main()
