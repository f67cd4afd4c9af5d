"""what would the 40^2 level cost on conv_xw with ragged 16 x 16 tiles (3 x 3 tiles per image, 69 % full)?  The same tile count at 48^2
(full tiles), timed by rocprofv3 / HIP events: against the quad kernel's 209 us (192 -> 192) and 378 us (384 -> 192) at 40^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conditional_score_diffusion_amd import ops
dev = torch.device('cuda:0')
for (C0, C1, Cout, res) in ((192, 0, 192, False), (192, 0, 192, True), (192, 192, 192, False), (96, 0, 192, False), (192, 96, 192, False)):
    B, H = 64, 48
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (Cin * 9)) ** 0.5; b = torch.randn(Cout, device=dev)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    r = torch.randn(B, H, H, Cout, device=dev) if res else None
    for _ in range(2):
        ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision='fp16x3', want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision='fp16x3', want_stats=True)
    e1.record(); torch.cuda.synchronize()
    print('C %d+%d -> %d @48 res=%d: %.1f us per call (incl. the weight pack)' % (C0, C1, Cout, res, e0.elapsed_time(e1) * 100))
