"""tuning aid: the weight gradient of one convolution shape, repeated (for rocprofv3 PMC / timing)
usage: python tools/wgrad_probe.py B Cin Cout H [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conditional_score_diffusion_amd import grad_ops_nhwc as G
B, Cin, Cout, H = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
layout = int(sys.argv[6]) if len(sys.argv) > 6 else 3      # 3 = NHWC operands, fp32 MFMA; 7 = split-bf16 MFMA
dev = torch.device('cuda:0')
x = torch.randn(B, H, H, Cin, device=dev)
w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).requires_grad_(True)
dy = torch.randn(B, H, H, Cout, device=dev)
from conditional_score_diffusion_amd import ops
from conditional_score_diffusion_amd._lib import lib, ptr, check, current_stream
dw = torch.empty_like(w)
sc = ops._scratch(lib().csd_conv_wgrad_scratch_bytes(B, Cin, Cout, H, H, 3, 1, 0), dev)
def run():
    check(lib().csd_conv2d_wgrad_ex(ptr(x), ptr(dy), ptr(dw), B, Cin, Cout, H, H, 3, 1, 0, 0, layout, ptr(sc), current_stream(dev)), 'wgrad')
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * B * H * H * Cin * Cout * 9
ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), w.shape, dy.permute(0, 3, 1, 2), padding=1) if B * H * H * Cin * Cout < 3e9 else None
if ref is not None: print('max rel err vs torch', float((dw - ref).abs().max() / ref.abs().max()))
print('wgrad B=%d %d->%d %dx%d: %.1f us  %.1f TFLOP/s' % (B, Cin, Cout, H, H, dt * 1e6, fl / dt / 1e12))
