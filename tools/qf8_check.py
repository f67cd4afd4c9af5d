import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'oracle'))
import numpy as np, torch
import cases, score_oracle as so
from conditional_score_diffusion_amd.models import utils as mutils
import conditional_score_diffusion_amd.models.ddpm
dev = torch.device('cuda:0')
def run(S, B, prec, ch_mult=(1, 1, 2, 2, 3, 3), attn=(20, 10, 5)):
    kw = dict(cases.SR3_160); kw['image_size'] = S; kw['ch_mult'] = ch_mult; kw['attn_resolutions'] = attn
    cfg = cases.make_config(**kw)
    cfg.model.csd_precision = prec
    model = mutils.create_model(cfg)
    nc = so.NetCfg.from_config(cfg)
    model.load_state_dict(so.synth_params(so.ddpm_param_shapes(nc), 0))
    model = model.to(dev).eval()
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32)).to(dev) * 20
    y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32)).to(dev)
    lab = torch.full((B,), 600., device=dev)
    with torch.no_grad():
        return model({'x': x, 'y': y}, lab).double().cpu()
def errs(a, b):
    rms = float(b.pow(2).mean().sqrt())
    return float((a - b).norm() / b.norm()), float(((a - b).abs() / (b.abs() + rms)).max())
for S, B, cm, at in ((160, 2, (1, 1, 2, 2, 3, 3), (20, 10, 5)), (40, 3, (1, 2, 2), (20, 10)), (80, 2, (1, 2, 3), (20,)), (32, 5, (1, 1, 2, 2, 3, 3), (4, 2, 1)), (64, 2, (1, 2, 2, 3), (8,))):
    ref = run(S, B, 'fp32', cm, at)
    for prec in ('fp16x3', 'fp16f8', 'fp16'):
        print(S, B, cm, prec, '%.3e %.3e' % errs(run(S, B, prec, cm, at), ref), flush=True)
