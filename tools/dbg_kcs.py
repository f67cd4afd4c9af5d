import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np, torch
import cases, score_oracle as so
from conditional_score_diffusion_amd.models import utils as mutils
dev = torch.device('cuda:0')
prec = sys.argv[1]
cfg = cases.make_config(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1), num_res_blocks=1, attn_resolutions=(), image_size=32)
cfg.model.csd_precision = prec
nc = so.NetCfg.from_config(cfg)
p = so.synth_params(so.ddpm_param_shapes(nc), 0)
m = mutils.create_model(cfg); m.load_state_dict(p); m = m.to(dev).eval()
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((2, 3, 32, 32)).astype(np.float32)); y = torch.from_numpy(rs.uniform(0, 1, (2, 3, 32, 32)).astype(np.float32))
lab = torch.ones(2) * 500
with torch.no_grad():
    ref = so.paired_forward(p, nc, x, y, lab, True)
    out = m({'x': x.to(dev), 'y': y.to(dev)}, lab.to(dev)).cpu()
err = (out - ref).abs()
print(prec, 'KCS', os.environ.get('CSD_FORCE_KCS'), 'rel err %.3e' % float(err.max() / ref.abs().max()), 'argmax', np.unravel_index(int(err.argmax()), err.shape))
e2 = err.amax(dim=(0, 1))
print('rows with err>1e-3*max:', (e2.amax(1) > 1e-3 * float(ref.abs().max())).nonzero().flatten().tolist()[:40])
print('cols with err>1e-3*max:', (e2.amax(0) > 1e-3 * float(ref.abs().max())).nonzero().flatten().tolist()[:40])
