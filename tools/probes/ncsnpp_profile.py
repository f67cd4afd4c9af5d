import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tools'))
import torch
import bench, bench_other
from conditional_score_diffusion_amd import _lib
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
cfg = bench_other.ncsnpp_config('ncsnpp_paired')
cfg.model.csd_precision = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
model = mutils.create_model(cfg)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(bench.synth_weights(shapes, 1)); model = model.to(dev).eval()
B = 64
x = torch.randn(B, 3, 160, 160, device=dev); y = torch.rand(B, 3, 160, 160, device=dev); lab = torch.full((B,), 500., device=dev)
with torch.no_grad():
    model({'x': x, 'y': y}, lab)
    _lib.profile_select(None, 1); _lib.profile_start()
    model({'x': x, 'y': y}, lab)
    p = _lib.profile_stop()
print({k: (round(v['ms'], 2), v['launches']) for k, v in p.items()})
