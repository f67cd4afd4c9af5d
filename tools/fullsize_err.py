import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import cases, score_oracle as so
from test_gpu_network import build, dev, rel
for prec in ['fp32','fp16x3','fp16']:
    cfg = cases.make_config(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), attn_resolutions=(20, 10, 5), image_size=160)
    cfg, nc, p, model = build(cfg, prec)
    rs = np.random.RandomState(5)
    B = 2
    lr = rs.uniform(0, 1, size=(B, 3, 20, 20)).astype(np.float32)
    y = torch.from_numpy(np.repeat(np.repeat(lr, 8, axis=2), 8, axis=3))
    x = torch.from_numpy((rs.standard_normal((B, 3, 160, 160)) * 3 + 0.5).astype(np.float32))
    labels = torch.tensor([300., 870.])
    with torch.no_grad():
        ref = so.paired_forward(p, nc, x, y, labels, sr3=True)
        out = model({'x': x.to(dev()), 'y': y.to(dev())}, labels.to(dev()))
    print(prec, 'B=2 rel err', rel(out.cpu().numpy(), ref.numpy()))
