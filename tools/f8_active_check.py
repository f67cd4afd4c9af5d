"""Is the fp8-correction path really taken?  Same network in fp16x3 and fp16f8: outputs must differ, errors vs the fp32 mode printed."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import cases
from test_gpu_network import build, dev
for S, B, cm, at in ((40, 3, (1, 2, 2), (20, 10)), (80, 2, (1, 2, 3), (20,)), (160, 1, (1, 1, 2, 2, 3, 3), (20, 10, 5))):
    outs = {}
    for prec in ('fp32', 'fp16x3', 'fp16f8'):
        kw = dict(cases.SR3_160); kw.update(image_size=S, ch_mult=cm, attn_resolutions=at)
        cfg = cases.make_config(**kw)
        cfg, nc, p, model = build(cfg, prec)
        rs = np.random.RandomState(3)
        x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32) * 20).to(dev())
        y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32)).to(dev())
        with torch.no_grad():
            outs[prec] = model({'x': x, 'y': y}, torch.full((B,), 600., device=dev())).double().cpu()
        print(S, prec, model.precision, model._cfg.precision)
    r = outs['fp32']
    for prec in ('fp16x3', 'fp16f8'):
        print('  ', S, prec, 'vs fp32 %.3e' % float((outs[prec] - r).norm() / r.norm()), 'identical to fp16x3:', bool(torch.equal(outs[prec], outs['fp16x3'])))
