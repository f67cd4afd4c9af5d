"""Side measurements for DESIGN.md (not the driver's bench): BASELINE configs[2] (CMDE inpainting 128x128, `ddpm_paired`,
two SDEs, fused PC loop) and the operator-granular NCSN++ executor (forward only).  One JSON line each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import cases, score_oracle as so
from conditional_score_diffusion_amd import sde_lib
from conditional_score_diffusion_amd.models import utils as mutils
from conditional_score_diffusion_amd.sampling import conditional, correctors, predictors

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16f8'


def cmde128(B=64, steps=20):
    cfg = cases.make_config(name='ddpm_paired', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), num_res_blocks=2, attn_resolutions=(16, 8, 4),
                            image_size=128, x_ch=3, y_ch=3, sigma_min_x=5e-3, sigma_max_x=float(np.sqrt(3 * 128 * 128)),
                            sigma_min_y=5e-3, sigma_max_y=1.0, snr=0.15)
    cfg.model.csd_precision = prec
    model = mutils.create_model(cfg)
    model.load_state_dict(so.synth_params(so.ddpm_param_shapes(so.NetCfg.from_config(cfg)), 0))
    model = model.to(dev).eval()
    sde = {'x': sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 1000), 'y': sde_lib.VESDE(5e-3, 1.0, 1000)}
    rs = np.random.RandomState(1)
    y = rs.uniform(0, 1, size=(B, 3, 128, 128)).astype(np.float32)
    y[:, :, 32:96, 32:96] = 0.
    y = torch.from_numpy(y).to(dev)
    # the fused loop with 3 and with `steps` PC steps: the difference isolates the per-step time (setup, prior, packing excluded)
    from conditional_score_diffusion_amd.sampling import fused
    res = {}
    for n in (3, steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused.run(model, sde, (B, 3, 128, 128), y, n, 0.15, 1e-5, True, seed=1)
        torch.cuda.synchronize(); res[n] = time.perf_counter() - t0
    t_step = (res[steps] - res[3]) / (steps - 3)
    print(json.dumps({'workload': 'BASELINE configs[2]: CMDE inpainting 128x128, ddpm_paired 6->6, two VE SDEs, fused PC loop',
                      'precision': prec, 'batch': B, 'ms_per_pc_step': t_step * 1e3, 'images_per_sec_1000_steps': B / (1000 * t_step)}))


def ncsnpp(name='ncsnpp_paired', B=8, reps=3):
    cfg = cases.make_ncsnpp_config(name=name, channels=6, nf=96, ch_mult=(1, 1, 2, 2, 3, 3), num_res_blocks=2,
                                   attn_resolutions=(20, 10, 5), image_size=160, embedding_type='positional')
    cfg.model.csd_precision = prec
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(cases.ncsnpp_params(shapes, 1))
    model = model.to(dev).eval()
    x = torch.randn(B, 3, 160, 160, device=dev)
    y = torch.rand(B, 3, 160, 160, device=dev)
    lab = torch.full((B,), 500., device=dev)
    with torch.no_grad():
        model({'x': x, 'y': y}, lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model({'x': x, 'y': y}, lab)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(json.dumps({'workload': 'NCSN++ (%s) with the SR3-160 hyper-parameters, forward only' % name,
                      'precision': prec, 'batch': B, 'ms_per_forward': dt * 1e3, 'images_per_sec_per_nfe': B / dt,
                      'params': sum(v.numel() for v in model.state_dict().values())}))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == 'bench':      # the two side figures of bench.py's line (B = 64, planned executors only)
        cmde128()
        ncsnpp('ncsnpp_paired', 64)
    else:
        cmde128()
        ncsnpp('ncsnpp_paired', 8)
        ncsnpp('ncsnpp_paired_ops', 8)
        ncsnpp('ncsnpp_paired', 64)
