"""bring-up aid for conv_xw: the shape sweep of tests/test_gpu_ops.py, one line per case BEFORE it runs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import torch
import test_gpu_ops as T
rs = np.random.RandomState(2024)
shapes = [(1, 16, 16), (1, 16, 32), (3, 16, 16), (1, 48, 16), (2, 32, 48), (5, 16, 32), (1, 80, 80), (7, 32, 32), (2, 64, 96)]
chans = [(64, 0, 64), (96, 0, 96), (64, 64, 128), (96, 96, 96), (128, 0, 192), (32, 96, 64), (160, 32, 288), (128, 128, 256),
         (80, 0, 96), (32, 0, 64), (48, 16, 192)]
for (B, H, W) in shapes:
    for k in rs.choice(len(chans), size=4, replace=False):
        C0, C1, Cout = chans[k]
        norm, res = bool(rs.randint(2)), bool(rs.randint(2))
        print(B, H, W, C0, C1, Cout, norm, res, flush=True)
        err, serr = T._block_case(rs, B, C0, C1, Cout, H, W, norm, res, 'fp16x3')
        torch.cuda.synchronize()
        print('   err %.2e stats %.2e' % (err, serr), flush=True)
