"""Tuning aid: per-workgroup phase timestamps of the fp16-source conv kernel (needs a library built with
-DCSD_C16_TIMING, selected through CSD_LIB_PATH).  Prints mean cycles per phase for the 160x160 layers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from conditional_score_diffusion_amd import _lib
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
cfg = bench.sr3_160_config(); cfg.model.csd_precision = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
B = 64
m = mutils.create_model(cfg); m.load_state_dict(bench.synth_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0)); m = m.to(dev).eval()
x = torch.randn(B, 3, 160, 160, device=dev) * 50; y = bench.synth_y(B).to(dev); lab = torch.full((B,), 500., device=dev)
L = 40
buf = torch.zeros(L * 4096 * 8, dtype=torch.int64, device=dev)
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12800
_lib.lib().csd_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
with torch.no_grad():
    m({'x': x, 'y': y}, lab)
    _lib.lib().csd_debug_timing(ctypes.c_void_p(buf.data_ptr()), nblocks, L)
    m({'x': x, 'y': y}, lab)
torch.cuda.synchronize()
allt = buf.cpu().numpy().reshape(L, 4096, 8)
print('%-28s %7s | %7s %7s %7s %7s %7s' % ('layer', 'total', 'tables', 'stg0ld', 'stg0mma', 'restK', 'epilog'))
for l in range(L):
    hdr = allt[l, 4095, :4]
    t = allt[l, :4095]
    t = t[t[:, 0] != 0]
    if len(t) == 0: continue
    d = np.diff(t[:, :6], axis=1).mean(0)
    print('#%2d Cin %3d Cout %3d in16 %d res %d temb %d %7.0f | %7.0f %7.0f %7.0f %7.0f %7.0f' % (
        l, hdr[0], hdr[1], hdr[2], hdr[3] & 1, (hdr[3] >> 1) & 1, (t[:, 5] - t[:, 0]).mean(), *d))
