"""Tuning aid: how busy are the workgroup slots of conv_ff_kernel?  Wall-clock stamps (100 MHz, common to all CUs) at the start and
end of every sampled workgroup: slot occupancy = sum of workgroup lifetimes / (kernel span x 512 slots).
   CSD_LIB_PATH=.../libcsd_hip_tune.so python tools/probes/ff_gap.py [precision] [shape index]"""
import ctypes, os, sys
os.environ['CSD_FF_ABL'] = str(int(os.environ.get('CSD_FF_ABL', '0')) | 128)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conditional_score_diffusion_amd import _lib, ops
import ff_probe
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16f8'
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
t = buf.cpu().numpy()[:4096 * 2 * 16].reshape(4096, 2, 16)[:, 0]
t = t[t[:, 14] != 0]
st, en = t[:, 14].astype(np.float64) * 0.01, t[:, 15].astype(np.float64) * 0.01       # us
nwg = B * (H // 16) ** 2 * (Cout // 96)
span = en.max() - st.min()
dur = en - st
print('%s: %d of %d workgroups sampled; lifetime mean %.1f us (p10 %.1f, p90 %.1f); sampled span %.1f us' %
      (prec, len(t), nwg, dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90), span))
# concurrency over time among the sampled workgroups (they are the first 4096 block ids = the first 64 % of the schedule)
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
tt = ev[:, 0] - st.min()
for lo, hi in ((0.1, 0.2), (0.2, 0.3), (0.3, 0.4), (0.4, 0.5)):
    m = (tt > lo * span) & (tt < hi * span)
    w_ = np.diff(tt[m], append=tt[m][-1])
    print('   mean concurrency in [%.0f %%, %.0f %%] of the span: %.0f of 512 slots' % (lo * 100, hi * 100, (conc[m] * w_).sum() / max(w_.sum(), 1e-9)))
order = np.argsort(st)
print('   start times of the 513th..520th workgroup: %s us; first 8 ends: %s' % (np.round(st[order][512:520] - st.min(), 1), np.round(np.sort(en)[:8] - st.min(), 1)))
