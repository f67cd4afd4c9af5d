"""bring-up aid for conv_xp: error structure of csd_conv3x3_block against fp64 torch on small cases"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conditional_score_diffusion_amd import ops
torch.manual_seed(0)
d = torch.device('cuda:0')


def run(B, C0, C1, Cout, H, W, norm, res, tag):
    Cin = C0 + C1
    x = torch.randn(B, H, W, Cin)
    w = torch.randn(Cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)
    bias = torch.randn(Cout)
    sc = torch.rand(B, Cin) + 0.5 if norm else None
    sh = torch.randn(B, Cin) * 0.5 if norm else None
    rv = torch.randn(B, H, W, Cout) if res else None
    o = lambda t: None if t is None else t.to(d)
    y, st = ops.conv3x3_block(x[..., :C0].contiguous().to(d), w.to(d), bias.to(d), x1=o(x[..., C0:].contiguous()) if C1 else None,
                              nscale=o(sc), nshift=o(sh), res=o(rv), precision='fp16x3', want_stats=True)
    torch.cuda.synchronize()
    xd = x.double()
    if norm:
        xd = torch.nn.functional.silu(xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :])
    ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if rv is not None:
        ref = ref + rv.double()
    e = (y.cpu().double() - ref).abs() / ref.abs().max()
    print('%s: max err %.3e' % (tag, e.max().item()))
    if e.max() > 1e-4:
        print('  per sample   ', [('%.1e' % v) for v in e.amax((1, 2, 3)).tolist()])
        print('  per row      ', [('%.1e' % v) for v in e.amax((0, 2, 3)).tolist()])
        print('  per col      ', [('%.1e' % v) for v in e.amax((0, 1, 3)).tolist()])
        print('  per cout/8   ', [('%.1e' % v) for v in e.amax((0, 1, 2)).reshape(-1, 8).amax(1).tolist()])
    yt = y.cpu().double().reshape(B, H // 16, 16, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(-1, 256, Cout)
    se = (st.cpu()[:, :, 0] - yt.sum(1)).abs().max().item() / yt.abs().sum(1).max().item()
    print('  stats err %.2e' % se)


run(1, 96, 0, 96, 16, 16, False, False, '1 tile raw')
run(1, 96, 0, 96, 16, 16, True, False, '1 tile norm')
run(1, 96, 0, 96, 16, 16, True, True, '1 tile norm res')
run(1, 96, 0, 96, 32, 32, True, False, '4 tiles norm')
run(2, 96, 96, 96, 16, 48, True, True, 'concat')
run(1, 64, 32, 192, 32, 16, True, True, '2 groups')
run(8, 96, 0, 96, 160, 160, True, True, 'big')
run(1, 64, 0, 64, 16, 16, False, False, 'nt2 1 tile raw')
run(1, 64, 0, 64, 16, 16, True, False, 'nt2 1 tile norm')
run(1, 128, 0, 128, 16, 16, True, False, 'nt2 2 groups')
run(2, 128, 0, 128, 32, 32, True, True, 'nt2 tiles res')
