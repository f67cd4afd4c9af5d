"""sanity: one complete 1000-step PC sampling run (BASELINE configs[1] network, random-init weights) through the public API"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from conditional_score_diffusion_amd import sde_lib
from conditional_score_diffusion_amd.models import utils as mutils
from conditional_score_diffusion_amd.sampling.conditional import get_conditional_sampling_fn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = bench.sr3_160_config()
cfg.model.csd_precision = sys.argv[2] if len(sys.argv) > 2 else 'fp16x3'
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = mutils.create_model(cfg)
model.load_state_dict(bench.synth_weights({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0))   # non-degenerate (bench.py)
model = model.to(dev).eval()
sde = sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
fn = get_conditional_sampling_fn(cfg, sde, [B] + list(cfg.data.shape_x), eps=1e-5)
lr = torch.rand(B, 3, 20, 20)
y = lr.repeat_interleave(8, 2).repeat_interleave(8, 3).to(dev)
torch.cuda.synchronize(); t0 = time.time()
x, info = fn(model, y)
torch.cuda.synchronize(); dt = time.time() - t0
print('B=%d %s: 1000-step PC in %.2f s (%.3f img/s); finite=%s min %.3f max %.3f mean %.3f' %
      (B, cfg.model.csd_precision, dt, B / dt, bool(torch.isfinite(x).all()), float(x.min()), float(x.max()), float(x.mean())))
