"""Tuning aid: per-tile cycle / wall stamps of conv_xk_kernel (tuning build: CSD_FF_ABL bit 7 + csd_debug_ff_timing).  Stamps sit at
tile boundaries only (see conv_xk.hip: a stamp inside the stream is a control-flow edge with live accumulators).
   CSD_LIB_PATH=.../libcsd_hip_tune.so python tools/xk_timing.py [shape index of tools/ff_probe.py]"""
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
t = t[t[:, 26] != 0]
print('shape C %d+%d -> %d @%d res=%d: %d workgroups sampled (their second tile)' % (C0, C1, Cout, H, res, len(t)))
ghz = ((t[:, 27] - t[:, 26]) / ((t[:, 29] - t[:, 28]) * 10.0)).mean()
print('one tile (%d stages, MFMA floor %d cycles): %.0f cycles in %.2f us -> shader clock %.3f GHz' % (
    Cin // 16, Cin // 16 * 3456, (t[:, 27] - t[:, 26]).mean(), (t[:, 29] - t[:, 28]).mean() * 0.01, ghz))
print('kernel wall per workgroup %.1f us' % ((t[:, 31] - t[:, 30]).mean() * 10.0 / 1e3))
