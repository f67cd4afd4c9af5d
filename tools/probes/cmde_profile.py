"""tuning aid: a few fused PC steps of the CMDE-128 side bench (BASELINE configs[2] shape, B = 64) - the target of tools/probes/timeline_cmde.sh"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tools'))
import numpy as np, torch
import bench_other
from conditional_score_diffusion_amd import sde_lib
from conditional_score_diffusion_amd.sampling import fused
bench_other.prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16f8'
cfg = bench_other.cmde128_config()
model = bench_other.build(cfg)
sde = {'x': sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 1000), 'y': sde_lib.VESDE(5e-3, 1.0, 1000)}
B = 64
y = torch.from_numpy(np.random.RandomState(1).uniform(0, 1, size=(B, 3, 128, 128)).astype(np.float32)).to(bench_other.dev)
fused.run(model, sde, (B, 3, 128, 128), y, 3, 0.15, 1e-5, True, seed=1)
torch.cuda.synchronize()
