"""tuning aid: per-class milliseconds of one NCSN++-256 (BASELINE configs[4] shape) evaluation at batch B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench, bench_other
from conditional_score_diffusion_amd import _lib
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
cfg = bench_other.ncsnpp_config('ncsnpp')
cfg.data.image_size = cfg.data.effective_image_size = 256
cfg.data.num_channels = 3
cfg.model.nf, cfg.model.ch_mult, cfg.model.attn_resolutions, cfg.model.embedding_type = 128, (1, 1, 2, 2, 2, 2, 2), (16,), 'fourier'
cfg.model.csd_precision = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = mutils.create_model(cfg)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(bench.synth_weights(shapes, 1)); model = model.to(dev).eval()
x = torch.randn(B, 3, 256, 256, device=dev); lab = torch.full((B,), 1.0, device=dev)
with torch.no_grad():
    model(x, lab)
    _lib.profile_select(None, 1); _lib.profile_start()
    model(x, lab)
    p = _lib.profile_stop()
tot = sum(v['ms'] for v in p.values())
print('B=%d %s total %.2f ms -> %.1f img*NFE/s' % (B, cfg.model.csd_precision, tot, B / tot * 1e3))
print({k: (round(v['ms'], 2), v['launches'], round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1)) for k, v in p.items()})
