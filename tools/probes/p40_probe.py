import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conditional_score_diffusion_amd import ops
dev = torch.device('cuda:0')
for (B, C0, C1, Cout, H, res) in [(16,192,0,192,80,True),(16,192,0,192,80,False),(16,192,192,192,80,False),(16,192,96,192,80,False),(16,96,0,192,80,False)]:
    Cin=C0+C1
    x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (Cin * 9)) ** 0.5; b = torch.randn(Cout, device=dev)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    r = torch.randn(B, H, H, Cout, device=dev) if res else None
    for prec in ['fp16f8']:
        f = lambda: ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision=prec, want_stats=True)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(prec, C0, C1, Cout, H, res, '%.1f us' % (e0.elapsed_time(e1)/20*1e3))
