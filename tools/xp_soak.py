"""Soak test of conv_xp (asm matrix instructions, no compiler-inserted wait states): thousands of launches per shape on a hot chip,
every result compared bitwise with the first one.     python tools/xp_soak.py [launches per shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conditional_score_diffusion_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
d = torch.device('cuda:0')
g = torch.Generator().manual_seed(11)
bad = 0
for (B, H, C0, C1, Cout, norm, res) in [(64, 160, 96, 0, 96, True, True), (64, 80, 192, 96, 96, True, False), (64, 64, 128, 0, 128, True, True),
                                        (50, 64, 128, 128, 128, False, False), (8, 160, 64, 64, 192, True, True)]:
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, generator=g).to(d)
    x1 = torch.randn(B, H, H, C1, generator=g).to(d) if C1 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(d)
    b = torch.randn(Cout, generator=g).to(d)
    sc = (torch.rand(B, Cin, generator=g) + 0.5).to(d) if norm else None
    sh = torch.randn(B, Cin, generator=g).to(d) if norm else None
    rv = torch.randn(B, H, H, Cout, generator=g).to(d) if res else None
    y0, s0 = ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=rv, precision='fp16x3', want_stats=True)
    y0, s0 = y0.clone(), s0.clone()
    t0 = time.time()
    diff = 0
    for i in range(n):
        y, s = ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=rv, precision='fp16x3', want_stats=True)
        if i % 25 == 24:                     # (the comparison itself is a kernel: every 25th launch keeps the chip on conv_xp)
            diff += int(not (torch.equal(y, y0) and torch.equal(s, s0)))
    torch.cuda.synchronize()
    bad += diff
    print('B %d %d^2 C %d+%d -> %d norm %d res %d: %d launches, %d checked, %d differ, %.1f s' %
          (B, H, C0, C1, Cout, norm, res, n, n // 25, diff, time.time() - t0))
print('SOAK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
