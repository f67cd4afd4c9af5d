"""Tuning aid: cycle stamps of conv_xw_kernel (tuning build: CSD_FF_ABL bit 7 + csd_debug_ff_timing).
   CSD_LIB_PATH=.../libcsd_hip_tune.so python tools/xw_timing.py [shape index of tools/ff_probe.py]"""
import ctypes, os, sys
os.environ['CSD_FF_ABL'] = str(int(os.environ.get('CSD_FF_ABL', '0')) | 128)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conditional_score_diffusion_amd import _lib, ops
import ff_probe
B, C0, C1, Cout, H, res = ff_probe.SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
dev = torch.device('cuda:0')
Cin = C0 + C1
x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (Cin * 9)) ** 0.5; b = torch.randn(Cout, device=dev)
sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
r = torch.randn(B, H, H, Cout, device=dev) if res else None
buf = torch.zeros(4096 * 2 * 16 + 1024 * 2 * 64, dtype=torch.int64, device=dev)
_lib.lib().csd_debug_ff_timing.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision='fp16x3', want_stats=True)
_lib.lib().csd_debug_ff_timing(ctypes.c_void_p(buf.data_ptr()))
ops.conv3x3_block(x0, w, b, x1=x1, nscale=sc, nshift=sh, res=r, precision='fp16x3', want_stats=True)
torch.cuda.synchronize()
t = buf.cpu().numpy()[:256 * 32].reshape(256, 32)
t = t[t[:, 0] != 0]
NS = min(Cin // 16, 12)
st = np.concatenate([t[:, :NS], t[:, 13:14]], axis=1) if NS == Cin // 16 else t[:, :NS + 1]
d = np.diff(st, axis=1)
print('shape C %d+%d -> %d @%d res=%d: %d workgroups sampled (their second tile)' % (C0, C1, Cout, H, res, len(t)))
print('unit cycles (3 row taps; MFMA floor 3456), mean: ' + ' '.join('%6.0f' % v for v in d.mean(0)))
print('                                          max : ' + ' '.join('%6.0f' % v for v in d.max(0)))
pr = np.concatenate([t[:, 14:23], t[:, 3:4]], axis=1)
print('unit 2, the nine products (12 MFMAs = 384 cycles each; hi*lo hi*hi lo*hi per row tap): ' + ' '.join('%.0f' % v for v in np.diff(pr, axis=1).mean(0)))
print('epilogue: last tap -> first store %.0f, stores %.0f, statistics %.0f' % ((t[:, 23] - t[:, 13]).mean(), (t[:, 24] - t[:, 23]).mean(), (t[:, 25] - t[:, 24]).mean()))
ghz = ((t[:, 27] - t[:, 26]) / ((t[:, 29] - t[:, 28]) * 10.0)).mean()
print('one tile: %.0f cycles in %.2f us -> shader clock %.3f GHz' % ((t[:, 27] - t[:, 26]).mean(), (t[:, 29] - t[:, 28]).mean() * 0.01, ghz))
print('kernel wall per workgroup %.1f us' % ((t[:, 31] - t[:, 30]).mean() * 10.0 / 1e3))
