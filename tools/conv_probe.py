"""Micro-probe: run a few convolution shapes of the SR3-160 network through the C ABI so that
rocprofv3 (--kernel-trace / --pmc) can look at the conv kernel in isolation.
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/conv_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from conditional_score_diffusion_amd import ops  # noqa: E402

SHAPES = [  # B, Cin, Cout, H, ksize
    (64, 96, 96, 160, 3),
    (64, 192, 96, 160, 3),
    (64, 96, 96, 80, 3),
    (64, 192, 192, 40, 3),
    (64, 192, 192, 20, 3),
    (64, 288, 288, 10, 3),
    (64, 288, 288, 5, 3),
    (64, 192, 96, 160, 1),
]


def main():
    reps = int(os.environ.get('REPS', '3'))
    only = os.environ.get('ONLY')
    dev = torch.device('cuda:0')
    for i, (B, Cin, Cout, H, ks) in enumerate(SHAPES):
        if only is not None and str(i) not in only.split(','):
            continue
        x = torch.randn(B, Cin, H, H, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, device=dev) * (1.0 / (Cin * ks * ks)) ** 0.5
        b = torch.randn(Cout, device=dev)
        for _ in range(reps):
            y = ops.conv2d(x, w, b, precision=os.environ.get('PREC', 'fp32'))
        torch.cuda.synchronize()
        fl = 2.0 * B * H * H * Cout * Cin * ks * ks
        print('shape', (B, Cin, Cout, H, ks), 'GFLOP %.1f' % (fl / 1e9), float(y.abs().mean()))


if __name__ == '__main__':
    main()
