import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conditional_score_diffusion_amd import ops
dev = torch.device('cuda:0')
n = 64 * 160 * 160 * 96
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: y.copy_(x)); print('torch copy  %.1f us  %.2f TB/s' % (dt * 1e6, 2 * n * 4 / dt / 1e12))
dt = t(lambda: torch.add(x, 1.0, out=y)); print('torch add   %.1f us  %.2f TB/s' % (dt * 1e6, 2 * n * 4 / dt / 1e12))
xs = x.view(64, -1); sc = torch.ones(64, device=dev)
dt = t(lambda: ops.scale_rows(xs, sc)); print('scale_rows (alloc+kernel) %.1f us  %.2f TB/s' % (dt * 1e6, 2 * n * 4 / dt / 1e12))
x4 = x.view(64, 96, 160, 160)
g, b = torch.ones(96, device=dev), torch.zeros(96, device=dev)
dt = t(lambda: x.sum()); print('torch sum   %.1f us  %.2f TB/s' % (dt * 1e6, n * 4 / dt / 1e12))
