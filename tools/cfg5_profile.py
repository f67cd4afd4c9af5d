"""tuning aid: per-class milliseconds of one NCSN++-256 (BASELINE configs[4] shape) evaluation at batch B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch, cases
from conditional_score_diffusion_amd import _lib
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
cfg = cases.make_ncsnpp_config(name='ncsnpp', channels=3, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                               attn_resolutions=(16,), image_size=256, embedding_type='fourier')
cfg.model.csd_precision = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = mutils.create_model(cfg)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(cases.ncsnpp_params(shapes, 1)); model = model.to(dev).eval()
x = torch.randn(B, 3, 256, 256, device=dev); lab = torch.full((B,), 1.0, device=dev)
with torch.no_grad():
    model(x, lab)
    _lib.profile_select(None, 1); _lib.profile_start()
    model(x, lab)
    p = _lib.profile_stop()
tot = sum(v['ms'] for v in p.values())
print('B=%d %s total %.2f ms -> %.1f img*NFE/s' % (B, cfg.model.csd_precision, tot, B / tot * 1e3))
print({k: (round(v['ms'], 2), v['launches'], round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1)) for k, v in p.items()})
