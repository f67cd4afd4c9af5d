"""Tuning aid: per-workgroup phase stamps of conv_ff_kernel (CSD_FF_ABL bit 7 + csd_debug_ff_timing).
   python tools/probes/ff_timing.py [precision] [shape index of tools/ff_probe.py]"""
import ctypes, os, sys
os.environ['CSD_FF_ABL'] = str(int(os.environ.get('CSD_FF_ABL', '0')) | 128)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conditional_score_diffusion_amd import _lib, ops
import ff_probe
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B, C0, C1, Cout, H, res = ff_probe.SHAPES[int(sys.argv[2]) if len(sys.argv) > 2 else 1]
dev = torch.device('cuda:0')
Cin = C0 + C1
x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (Cin * 9)) ** 0.5; b = torch.randn(Cout, device=dev)
sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
r = torch.randn(B, H, H, Cout, device=dev) if res else None
buf = torch.zeros(4096 * 2 * 16 + 1024 * 2 * 64, dtype=torch.int64, device=dev)
_lib.lib().csd_debug_ff_timing.argtypes = [ctypes.c_void_p]
ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision=prec, want_stats=True)
_lib.lib().csd_debug_ff_timing(ctypes.c_void_p(buf.data_ptr()))
ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision=prec, want_stats=True)
torch.cuda.synchronize()
allb = buf.cpu().numpy()
t = allb[:4096 * 2 * 16].reshape(4096, 2, 16)
fine = allb[4096 * 2 * 16:].reshape(1024, 2, 64)
for wv in (0, 1):
    tt = t[:, wv]
    tt = tt[tt[:, 0] != 0]
    wall = (tt[:, 15] - tt[:, 14]).mean() * 10.0      # ns (wall_clock64: 100 MHz)
    tt = tt[:, :14]
    nz = int((tt[0] != 0).sum())
    d = np.diff(tt[:, :nz], axis=1)
    tot = (tt[:, nz - 1] - tt[:, 0]).mean()
    print('wave %d: %d stamps, wg total mean %.0f clk = %.1f us wall -> %.2f GHz | mean deltas: %s' % (wv, nz, tot, wall / 1e3, tot / wall, ' '.join('%6.0f' % v for v in d.mean(0))))
    # late workgroups (steady state: both slots of a CU busy) vs the first wave of workgroups
    late = tt[tt[:, 0] > np.percentile(tt[:, 0], 30)]
    dl = np.diff(late[:, :nz], axis=1)
    print('   steady-state: total %.0f | %s' % ((late[:, nz - 1] - late[:, 0]).mean(), ' '.join('%6.0f' % v for v in dl.mean(0))))
span = t[:, 0, :14][t[:, 0, 0] != 0]
print('kernel span (first start -> last end among sampled wgs): %.0f clk' % (span.max() - span[:, 0].min()))

if int(os.environ.get('CSD_FF_ABL', '0')) & 256:
    # per-step anatomy of stage 2 (split mode: one step = one ring group): [top .. first MFMA triple .. fragment reads issued .. 15 MFMAs
    # .. (wave 0: DMA wait / wave 1: conversion) .. barrier passed = next top]
    for wv in (0, 1):
        f = fine[:, wv]
        f = f[f[:, 0] != 0]
        n = int((f[0] != 0).sum())
        d = np.diff(f[:, :n], axis=1).mean(0)
        print('wave %d stage 2, per step [mma0 | issue reads | 15 mma | tail | barrier]:' % wv)
        for st in range(n // 5):
            seg = d[st * 5:st * 5 + 5]
            print('   step %d: %s' % (st, ' '.join('%6.0f' % v for v in seg)))
