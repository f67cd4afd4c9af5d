"""-m gpu: parity at the BENCHMARKED shapes.

* the full-size SR3-160 network (BASELINE configs[1]) over the REAL 1000-step PC schedule against a run of the imported
  reference itself (tests/golden/sr3_160_long.npz, oracle/make_goldens.py:gen_sr3_160_long) - in every precision mode,
  at the north-star tolerance 1e-3, norm-wise AND element-wise;
* batch 64 (the bench batch): sample 17 of a B = 64 evaluation equals the B = 1 evaluation bit for bit, and matches the oracle;
* BASELINE configs[4]'s network (NCSN++ 256 x 256, nf = 128, seven levels, attention at 16) against the oracle.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
import score_oracle as so  # noqa: E402

# CERTIFIED modes hold the north-star tolerance ("within 1e-3 rel fp32") norm-wise AND element-wise, at full size, over the
# whole 1000-step schedule.  The plain-fp16 mode does NOT: measured on these same tests 6.8e-4 .. 9.1e-4 norm-wise but
# 2.7e-3 .. 3.9e-3 element-wise (1.2e-3 / 4.5e-3 on NCSN++-256) - it is an optional throughput mode, never the bench default,
# and is held here to ITS OWN documented bound (FP16_TOL) so that a regression of it is still caught.
MODES = [('fp32', 1e-3, 1e-3), ('fp16x3', 1e-3, 1e-3), ('fp16f8', 1e-3, 1e-3), ('fp16', 2e-3, 6e-3)]
TOL = 1e-3
FP16_TOL = (2e-3, 6e-3)      # (norm-wise, element-wise): NOT the north-star tolerance


def dev():
    return torch.device('cuda:0')


def normwise(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()


def elementwise(a, b):
    """max over elements of |a - b| / (|b| + rms(b)): <= tol  <=>  |a - b| <= tol |b| + tol rms(b) everywhere"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (np.abs(a - b) / (np.abs(b) + np.sqrt((b * b).mean()))).max()


def build_sr3_160(precision):
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.make_config(**cases.SR3_160)
    cfg.model.csd_precision = precision
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    model = mutils.create_model(cfg)
    model.load_state_dict(p)
    return cfg, nc, p, model.to(dev()).eval()


_long = {}


def long_inputs():
    if not _long:
        _long['tape'] = cases.long_tape()
        _long['y'] = cases.sr3_160_y(cases.LONG_B)
    return _long['tape'], _long['y']


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES)
def test_long_schedule_vs_reference(golden_dir, precision, tol_n, tol_e):
    """1000 PC steps (2000 network evaluations), B = 2, seeded noise tape, fused device loop; x after every 50th step and the
    final denoised sample against the reference's own run"""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import fused
    g = np.load(os.path.join(golden_dir, 'sr3_160_long.npz'))
    cfg, nc, p, model = build_sr3_160(precision)
    sde = sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    tape, y = long_inputs()
    P, B, st = cases.LONG_P, cases.LONG_B, cases.LONG_STRIDE
    x, rec, _ = fused.run(model, sde, (B, 3, 160, 160), y.to(dev()), P, cfg.sampling.snr, 1e-5, True, noise_tape=tape, record=True)
    worst_n = worst_e = 0.0
    for j, step in enumerate(g['steps']):
        got = rec[int(step)][:, :, ::st, ::st].cpu().numpy()
        worst_n = max(worst_n, normwise(got, g['evo'][j]))
        worst_e = max(worst_e, elementwise(got, g['evo'][j]))
    fin_n, fin_e = normwise(x.cpu().numpy(), g['final']), elementwise(x.cpu().numpy(), g['final'])
    print('long schedule %s: snapshots norm-wise %.3e element-wise %.3e; final %.3e / %.3e'
          % (precision, worst_n, worst_e, fin_n, fin_e))
    assert max(worst_n, fin_n) < tol_n and max(worst_e, fin_e) < tol_e


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES)
def test_full_size_forward_vs_reference(golden_dir, precision, tol_n, tol_e):
    """one evaluation of the full-size network against the reference's own output"""
    g = np.load(os.path.join(golden_dir, 'sr3_160_long.npz'))
    cfg, nc, p, model = build_sr3_160(precision)
    y = cases.sr3_160_y(cases.LONG_B)[:1]
    with torch.no_grad():
        out = model({'x': torch.from_numpy(g['fwd_x']).to(dev()), 'y': y.to(dev())}, torch.from_numpy(g['fwd_label']).to(dev()))
    n, e = normwise(out.cpu().numpy(), g['fwd_net']), elementwise(out.cpu().numpy(), g['fwd_net'])
    print('full-size forward %s: %.3e norm-wise, %.3e element-wise' % (precision, n, e))
    assert n < tol_n and e < tol_e


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES)
def test_batch_64_equals_batch_1(precision, tol_n, tol_e):
    """the bench batch: every tile schedule the B = 64 plan picks gives sample 17 the bits the B = 1 plan gives it, and the
    oracle's values"""
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cfg, nc, p, model = build_sr3_160(precision)
    B, k = 64, 17
    g = torch.Generator().manual_seed(64)
    lr = torch.rand(B, 3, 20, 20, generator=g)
    y = lr.repeat_interleave(8, dim=2).repeat_interleave(8, dim=3).contiguous()
    x = torch.randn(B, 3, 160, 160, generator=g) * 40.0 + 0.5
    lab = torch.full((B,), 611.0)
    with torch.no_grad():
        full = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev()))
        one = model({'x': x[k:k + 1].to(dev()), 'y': y[k:k + 1].to(dev())}, lab[:1].to(dev()))
        ref = so.paired_forward(p, nc, x[k:k + 1], y[k:k + 1], lab[:1], sr3=True)
    assert torch.equal(full[k:k + 1], one)
    n, e = normwise(full[k:k + 1].cpu().numpy(), ref.numpy()), elementwise(full[k:k + 1].cpu().numpy(), ref.numpy())
    print('B=64 sample %d vs oracle, %s: %.3e / %.3e' % (k, precision, n, e))
    assert n < tol_n and e < tol_e


def build_ncsnpp_256(precision):
    """BASELINE configs[4]: NCSN++ at 256 x 256, nf = 128, ch_mult (1,1,2,2,2,2,2), attention at 16, Fourier embedding,
    input/output pyramids, 65.57 M parameters (configs/ve/ffhq_256_ncsnpp_continuous.py; VESDE 0.01 - 348, N = 2000:
    configs/default_lsun_configs.py:51-53 + ffhq_256_ncsnpp_continuous.py:45)"""
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.make_ncsnpp_config(name='ncsnpp', channels=3, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                                   attn_resolutions=(16,), image_size=256, embedding_type='fourier')
    cfg.model.csd_precision = precision
    cfg.model.num_scales, cfg.model.sigma_min, cfg.model.sigma_max = 2000, 0.01, 348.
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sum(int(np.prod(s)) for s in shapes.values()) == 65574549     # 65.57 M
    p = cases.ncsnpp_params(shapes, 3)
    model.load_state_dict(p)
    return cfg, p, model.to(dev()).eval()


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES[1:])
def test_config5_ncsnpp_256_vs_oracle(precision, tol_n, tol_e):
    """BASELINE configs[4]'s network, one evaluation at B = 2 in the fp16-MFMA arithmetic modes"""
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cfg, p, model = build_ncsnpp_256(precision)
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.uniform(-1, 2, size=(2, 3, 256, 256)).astype(np.float32))
    labels = torch.tensor([np.log(3.7), np.log(0.05)], dtype=torch.float32)
    with torch.no_grad():
        got = model(x.to(dev()), labels.to(dev())).cpu()
        ref = so.ncsnpp_forward(p, cfg, x, labels)
    n, e = normwise(got.numpy(), ref.numpy()), elementwise(got.numpy(), ref.numpy())
    print('NCSN++ 256 %s: %.3e / %.3e' % (precision, n, e))
    assert n < tol_n and e < tol_e


_cache = {}


def _ncsnpp_256_pc_reference(P):
    """the oracle's unconditional PC loop (sampling/unconditional.py:194-226 restated) around oracle.ncsnpp_forward: computed once"""
    if ('nc256', P) not in _cache:
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        cfg, p, _ = build_ncsnpp_256('fp32')
        shape = (1, 3, 256, 256)
        tp = cases.tape([shape] * (1 + 2 * P), seed=256)
        ve = so.VE(0.01, 348., 2000)

        def score_fn(x, t):
            std = ve.std(t)
            return so.ncsnpp_forward(p, cfg, x, torch.log(std)) / std[:, None, None, None]

        with torch.no_grad():
            ref = so.pc_sample_unconditional(score_fn, shape, so.NoiseTape(tp), ve, p_steps=P, snr=0.075, eps=1e-5, denoise=True)
        _cache[('nc256', P)] = (tp, ref.numpy())
    return _cache[('nc256', P)]


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES[1:3])
def test_config5_ncsnpp_256_pc_steps_vs_oracle(precision, tol_n, tol_e):
    """BASELINE configs[4] as a SAMPLER: three fused unconditional PC steps (reverse diffusion + Langevin, snr 0.075, VESDE 0.01 - 348,
    N = 2000, Fourier labels = log sigma) of the 2000-step schedule on NCSN++-256 with a noise tape, against the oracle's loop"""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors, predictors, unconditional
    P = 3
    tp, ref = _ncsnpp_256_pc_reference(P)
    cfg, p, model = build_ncsnpp_256(precision)
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=348., N=2000)
    shape = (1, 3, 256, 256)
    fn = unconditional.get_pc_sampler(sde, shape, predictors.get_predictor('reverse_diffusion'), correctors.get_corrector('langevin'),
                                      snr=0.075, p_steps=P, c_steps=1, continuous=True, denoise=True, eps=1e-5)
    got, _ = fn(model, noise_tape=tp)
    n, e = normwise(got.cpu().numpy(), ref), elementwise(got.cpu().numpy(), ref)
    print('NCSN++ 256 %d PC steps %s: %.3e / %.3e' % (P, precision, n, e))
    assert n < tol_n and e < tol_e


# ---- BASELINE configs[2]: CMDE inpainting at 128 x 128 (configs/ve/inverse_problems/inpainting/celebA_ours_NDV.py:100-135:
#      ddpm_paired 6 -> 6 channels, nf 96, ch_mult (1,1,2,2,3,3), attention at 16 / 8 / 4, x: cVESDE(5e-3, sqrt(3*128^2)), y: VESDE(5e-3, 1)) ----
def build_cmde_128(precision):
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.make_config(name='ddpm_paired', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), attn_resolutions=(16, 8, 4), image_size=128,
                            sigma_min_y=5e-3, sigma_max_y=1.0)
    cfg.model.csd_precision = precision
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 2)
    model = mutils.create_model(cfg)
    model.load_state_dict(p)
    return cfg, nc, p, model.to(dev()).eval()


def _cmde_y(B, seed=128):
    """inpainting conditioning: the image with a 64 x 64 square (25 % of the area) zeroed (SRFLOWDataset.py:321-325)"""
    rs = np.random.RandomState(seed)
    y = rs.uniform(0, 1, size=(B, 3, 128, 128)).astype(np.float32)
    for b in range(B):
        r, c = rs.randint(0, 65, size=2)
        y[b, :, r:r + 64, c:c + 64] = 0.
    return torch.from_numpy(y)


def _cmde_128_reference(P):
    if ('cmde128', P) not in _cache:
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        cfg, nc, p, _ = build_cmde_128('fp32')
        B = 2
        y = _cmde_y(B)
        xs = (B, 3, 128, 128)
        tp = cases.tape([xs] + [xs, xs, xs, xs] * P, seed=129)        # prior | per step: z_y(corr), z_corr, z_y(pred), z_pred
        rs = np.random.RandomState(130)
        x = torch.from_numpy((rs.standard_normal(xs) * 30.0 + 0.5).astype(np.float32))
        lab = torch.tensor([611.0, 87.25])
        with torch.no_grad():
            fwd = so.paired_forward(p, nc, x, y, lab, sr3=False)
            ref = so.pc_sample_conditional(p, nc, y, so.NoiseTape(tp), (cfg.model.sigma_min_x, cfg.model.sigma_max_x),
                                           (cfg.model.sigma_min_y, cfg.model.sigma_max_y), sr3=False, p_steps=P, snr=cfg.sampling.snr,
                                           N=1000)
        _cache[('cmde128', P)] = (y, tp, x, lab, torch.cat([fwd['x'], fwd['y']], dim=1).numpy(), ref.numpy())
    return _cache[('cmde128', P)]


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES[:3])
def test_config3_cmde_128_vs_oracle(precision, tol_n, tol_e):
    """BASELINE configs[2] at ITS OWN size, B = 2: one evaluation of the 6 -> 6 channel network (both score halves) and three fused
    two-SDE PC steps (fresh y_t for the corrector and for the predictor: four draws per step) with a noise tape, against the oracle -
    the paired last layer's tap-partial kernel, conv_ff at 128 / 64, the quad kernel at 32 / 16 / 8 / 4 and L = 256 / 64 / 16
    attention with C = 192 / 288 together at that shape"""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import conditional
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    P = 3
    y, tp, x, lab, fwd_ref, pc_ref = _cmde_128_reference(P)
    cfg, nc, p, model = build_cmde_128(precision)
    with torch.no_grad():
        out = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev()))
    got = torch.cat([out['x'], out['y']], dim=1).cpu().numpy()
    n, e = normwise(got, fwd_ref), elementwise(got, fwd_ref)
    print('CMDE-128 forward %s: %.3e / %.3e' % (precision, n, e))
    assert n < tol_n and e < tol_e
    sde = {'x': sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 1000),
           'y': sde_lib.VESDE(cfg.model.sigma_min_y, cfg.model.sigma_max_y, 1000)}
    xs = tuple(x.shape)
    sampler = conditional.get_pc_conditional_sampler(sde, xs, get_predictor('conditional_reverse_diffusion'),
                                                     get_corrector('conditional_langevin'), snr=cfg.sampling.snr, p_steps=P,
                                                     c_steps=1, continuous=True, denoise=True, eps=1e-5)
    smp, _ = sampler(model, y.to(dev()), noise_tape=tp)
    n, e = normwise(smp.cpu().numpy(), pc_ref), elementwise(smp.cpu().numpy(), pc_ref)
    print('CMDE-128 %d PC steps %s: %.3e / %.3e' % (P, precision, n, e))
    assert n < tol_n and e < tol_e


@pytest.mark.parametrize('precision', ['fp16x3', 'fp16f8'])
def test_config3_cmde_128_batch_64_equals_batch_1(precision):
    """the side-bench batch of configs[2]: sample 17 of a B = 64 evaluation has the bits of the B = 1 evaluation"""
    cfg, nc, p, model = build_cmde_128(precision)
    B, k = 64, 17
    y = _cmde_y(B, seed=64)
    g = torch.Generator().manual_seed(65)
    x = torch.randn(B, 3, 128, 128, generator=g) * 40.0 + 0.5
    lab = torch.full((B,), 611.0)
    with torch.no_grad():
        full = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev()))
        one = model({'x': x[k:k + 1].to(dev()), 'y': y[k:k + 1].to(dev())}, lab[:1].to(dev()))
    assert torch.equal(full['x'][k:k + 1], one['x']) and torch.equal(full['y'][k:k + 1], one['y'])


def test_profiler_prices_the_dominant_kernel_on_survey_8d_bytes():
    """the in-library profiler's class 'conv3x3' is the conv_xk launches alone (30 per evaluation of the SR3-160 network; the first layer
    and the quad kernel's 3x3 launches report as 'conv3x3_other'), and csd_profile_stop_ex returns for it the SURVEY.md 8(d) bytes -
    (input + output tensor) x 4 B = 19.98 GB per evaluation at B = 64, i.e. 312.2 MB per image - next to the bytes the kernels have to
    move (larger: every block's second convolution also reads a residual).  bench.py's roofline.frac is computed from the former."""
    from conditional_score_diffusion_amd import _lib
    cfg, nc, p, model = build_sr3_160('fp16x3')
    B = 2
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, 3, 160, 160, generator=g) * 30).to(dev())
    y = cases.sr3_160_y(B).to(dev())
    lab = torch.full((B,), 500.0, device=dev())
    with torch.no_grad():
        model({'x': x, 'y': y}, lab)                     # (packs the weights, builds the plan)
        _lib.profile_select(None, 1)
        _lib.profile_start()
        model({'x': x, 'y': y}, lab)
        torch.cuda.synchronize()
        prof = _lib.profile_stop()
    dom, other = prof['conv3x3'], prof['conv3x3_other']
    assert dom['launches'] == 30 and other['launches'] > 20
    assert abs(dom['alg_bytes'] / B - 19.98e9 / 64) < 0.01 * 19.98e9 / 64
    assert dom['bytes'] > 1.15 * dom['alg_bytes']         # 15 of the 30 launches read a residual of their output's size
    assert dom['flops'] > 0.9 * (dom['flops'] + other['flops'])      # 92 % of the 3x3 stride-1 flops of this shape
    for k, v in prof.items():
        assert v['alg_bytes'] <= v['bytes'] * (1 + 1e-9) or k in ('conv3x3_other', 'conv3x3_resample', 'conv1x1'), k
