"""Side measurements for DESIGN.md (not the driver's bench; imports nothing from oracle/): BASELINE configs[2] (CMDE inpainting 128x128, `ddpm_paired`,
two SDEs, fused PC loop) and the operator-granular NCSN++ executor (forward only).  One JSON line each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench                                   # config / random-init helpers of the bench (nothing here touches oracle/)
from conditional_score_diffusion_amd import sde_lib
from conditional_score_diffusion_amd.config_dict import ConfigDict
from conditional_score_diffusion_amd.models import utils as mutils
import conditional_score_diffusion_amd.models.ddpm      # noqa: F401  (registers the model names)
import conditional_score_diffusion_amd.models.ncsnpp    # noqa: F401

dev = torch.device('cuda:0')
prec = sys.argv[1] if (len(sys.argv) > 1 and __name__ == '__main__') else 'fp16x3'


def cmde128_config():
    """BASELINE configs[2]: configs/ve/inverse_problems/inpainting/celebA_ours_DV_128-like values that the hot path reads"""
    c = bench.sr3_160_config()
    S = 128
    c.data = ConfigDict(image_size=S, effective_image_size=S, centered=False, shape_x=[3, S, S], shape_y=[3, S, S], num_channels=6)
    c.model.name = 'ddpm_paired'
    c.model.attn_resolutions = (16, 8, 4)
    c.model.sigma_max_x = float(np.sqrt(3 * S * S))
    c.model.sigma_max = c.model.sigma_max_x
    c.model.sigma_max_y = 1.0
    c.model.output_channels = 6
    c.sampling.snr = 0.15
    return c


def ncsnpp_config(name):
    c = ConfigDict()
    S = 160
    c.training = ConfigDict(continuous=True, sde='vesde', likelihood_weighting=False, reduce_mean=False)
    c.sampling = ConfigDict(method='pc', predictor='reverse_diffusion', corrector='langevin', n_steps_each=1, noise_removal=True,
                            probability_flow=False, snr=0.075)
    c.data = ConfigDict(image_size=S, effective_image_size=S, centered=False, num_channels=6, shape_x=[3, S, S], shape_y=[3, S, S])
    c.model = ConfigDict(name=name, nf=96, ch_mult=(1, 1, 2, 2, 3, 3), num_res_blocks=2, attn_resolutions=(20, 10, 5), dropout=0.1,
                         resamp_with_conv=True, conditional=True, nonlinearity='swish', num_scales=1000, sigma_min=0.01, sigma_max=50.,
                         fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan', progressive='output_skip',
                         progressive_input='input_skip', progressive_combine='sum', attention_type='ddpm', init_scale=0.,
                         embedding_type='positional', fourier_scale=16, conv_size=3, scale_by_sigma=True)
    return c


def build(cfg):
    cfg.model.csd_precision = prec
    model = mutils.create_model(cfg)
    model.load_state_dict(bench.synth_weights({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0))
    return model.to(dev).eval()


def cmde128(B=64, steps=20):
    cfg = cmde128_config()
    model = build(cfg)
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
    model = build(ncsnpp_config(name))
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


def ncsnpp256(B=8, reps=3):
    """BASELINE configs[4]: NCSN++ 256 x 256 (nf = 128, ch_mult (1,1,2,2,2,2,2), attention at 16, Fourier embedding, pyramids), forward only,
    at the per-GPU batch of an 8-GPU sampling run"""
    c = ncsnpp_config('ncsnpp')
    S = 256
    c.data = ConfigDict(image_size=S, effective_image_size=S, centered=False, num_channels=3)
    c.model.nf, c.model.ch_mult, c.model.attn_resolutions, c.model.embedding_type = 128, (1, 1, 2, 2, 2, 2, 2), (16,), 'fourier'
    c.model.num_scales, c.model.sigma_max = 2000, 348.
    model = build(c)
    x = torch.randn(B, 3, S, S, device=dev)
    lab = torch.full((B,), float(np.log(3.7)), device=dev)
    with torch.no_grad():
        model(x, lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model(x, lab)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(json.dumps({'workload': 'BASELINE configs[4]: NCSN++ 256x256 nf=128, forward only', 'precision': prec, 'batch': B,
                      'ms_per_forward': dt * 1e3, 'images_per_sec_per_nfe': B / dt,
                      'params': sum(v.numel() for v in model.state_dict().values())}))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == 'bench':      # the side figures of bench.py's line (planned executors only)
        cmde128()
        ncsnpp('ncsnpp_paired', 64)
        ncsnpp256(8)
        ncsnpp256(32)
    else:
        cmde128()
        ncsnpp('ncsnpp_paired', 8)
        ncsnpp('ncsnpp_paired_ops', 8)
        ncsnpp('ncsnpp_paired', 64)
