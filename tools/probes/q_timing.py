import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np, torch
import bench
from conditional_score_diffusion_amd import _lib
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
cfg = bench.sr3_160_config(); cfg.model.csd_precision = 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = mutils.create_model(cfg); m.load_state_dict(bench.synth_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0)); m = m.to(dev).eval()
x = torch.randn(B, 3, 160, 160, device=dev) * 50; y = bench.synth_y(B).to(dev); lab = torch.full((B,), 500., device=dev)
L = 12
buf = torch.zeros(L * 4096 * 8, dtype=torch.int64, device=dev)
nblocks = int(sys.argv[1])
_lib.lib().csd_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
with torch.no_grad():
    m({'x': x, 'y': y}, lab)
    _lib.lib().csd_debug_timing(ctypes.c_void_p(buf.data_ptr()), nblocks, L)
    m({'x': x, 'y': y}, lab)
torch.cuda.synchronize()
allt = buf.cpu().numpy().reshape(L, 4096, 8)
for l in range(L):
    hdr = allt[l, 4095, :4]
    t = allt[l, :4095]
    t = t[t[:, 0] != 0]
    if len(t) == 0: continue
    nz = int((t[0] != 0).sum())
    d = np.diff(t[:, :nz], axis=1).mean(0)
    print('#%2d nblocks %d Cin %3d Cout %3d res %d temb %d wg total %7.0f | ' % (l, nblocks, hdr[0], hdr[1], hdr[3] & 1, (hdr[3] >> 1) & 1, (t[:, nz - 1] - t[:, 0]).mean()) + ' '.join('%6.0f' % v for v in d), ' first wg start spread %.0f'%(t[:,0].max()-t[:,0].min()))
