"""ORACLE tooling: the seeded parity cases shared by oracle/make_goldens.py (which runs the
imported reference on them) and tests/ (which run the oracle and the HIP path on them).
Inputs are regenerated from seeds (numpy legacy RandomState: version-stable), so fixtures hold
only the reference's *outputs*."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conditional_score_diffusion_amd.config_dict import ConfigDict  # noqa: E402


def make_config(name='ddpm_paired_SR3', nf=32, ch_mult=(1, 2, 2), num_res_blocks=2,
                attn_resolutions=(10, 5), image_size=20, x_ch=3, y_ch=3, num_scales=1000,
                sigma_min_x=5e-3, sigma_max_x=None, sigma_min_y=5e-3, sigma_max_y=1.0, snr=0.15):
    """A reference-style config carrying exactly the keys the hot path reads
    (cf. configs/ve/inverse_problems/super_resolution/celebA_SR3_160.py:6-161)."""
    c = ConfigDict()
    c.training = ConfigDict(continuous=True, sde='vesde', likelihood_weighting=True, reduce_mean=True,
                            conditioning_approach='sr3' if name.endswith('SR3') else 'ours_NDV')
    c.sampling = ConfigDict(method='pc', predictor='conditional_reverse_diffusion',
                            corrector='conditional_langevin', n_steps_each=1, noise_removal=True,
                            probability_flow=False, snr=snr)
    if name == 'ddpm':
        c.sampling.predictor, c.sampling.corrector = 'reverse_diffusion', 'langevin'
    c.data = ConfigDict(image_size=image_size, effective_image_size=image_size, centered=False,
                        shape_x=[x_ch, image_size, image_size], shape_y=[y_ch, image_size, image_size],
                        num_channels=x_ch + y_ch)
    if sigma_max_x is None:
        sigma_max_x = float(np.sqrt(np.prod(c.data.shape_x)))
    paired = name != 'ddpm'
    c.model = ConfigDict(name=name, nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks,
                         attn_resolutions=tuple(attn_resolutions), dropout=0.1, resamp_with_conv=True,
                         conditional=True, nonlinearity='swish', num_scales=num_scales,
                         sigma_min_x=sigma_min_x, sigma_max_x=sigma_max_x,
                         sigma_min_y=sigma_min_y, sigma_max_y=sigma_max_y,
                         sigma_min=sigma_min_x, sigma_max=sigma_max_x,
                         input_channels=(x_ch + y_ch) if paired else x_ch,
                         output_channels=x_ch if name != 'ddpm_paired' else x_ch + y_ch,
                         embedding_type='positional', scale_by_sigma=True, ema_rate=0.999)
    # configs/ve/inverse_problems/super_resolution/celebA_SR3_160.py:149-157
    c.optim = ConfigDict(weight_decay=0, optimizer='Adam', lr=2e-4, beta1=0.9, eps=1e-8, warmup=2500, grad_clip=1)
    c.seed = 42
    return c


CASES = {
    # name: (config kwargs, batch)
    'sr3_tiny': (dict(name='ddpm_paired_SR3'), 2),
    'cmde_tiny': (dict(name='ddpm_paired'), 2),
    'uncond_tiny': (dict(name='ddpm', ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(8,),
                         image_size=16), 2),
}


def make_ncsnpp_config(name='ncsnpp', nf=16, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16,
                       channels=3, embedding_type='fourier', progressive='output_skip', progressive_input='input_skip',
                       skip_rescale=True, fir_kernel=(1, 3, 3, 1), init_scale=0., centered=False, fir=True):
    """A reference-style NCSN++ config with the keys models/ncsnpp.py:44-236 reads
    (cf. configs/ve/ffhq_256_ncsnpp_continuous.py, configs/default_lsun_configs.py)."""
    c = ConfigDict()
    c.training = ConfigDict(continuous=True, sde='vesde', likelihood_weighting=False, reduce_mean=False)
    c.sampling = ConfigDict(method='pc', predictor='reverse_diffusion', corrector='langevin', n_steps_each=1,
                            noise_removal=True, probability_flow=False, snr=0.075)
    c.data = ConfigDict(image_size=image_size, effective_image_size=image_size, centered=centered, num_channels=channels)
    c.model = ConfigDict(name=name, nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks,
                         attn_resolutions=tuple(attn_resolutions), dropout=0.1, resamp_with_conv=True, conditional=True,
                         nonlinearity='swish', num_scales=1000, sigma_min=0.01, sigma_max=50., fir=fir,
                         fir_kernel=list(fir_kernel), skip_rescale=skip_rescale, resblock_type='biggan',
                         progressive=progressive, progressive_input=progressive_input, progressive_combine='sum',
                         attention_type='ddpm', init_scale=init_scale, embedding_type=embedding_type, fourier_scale=16,
                         conv_size=3, scale_by_sigma=True)
    return c


NCSNPP_CASES = {
    # name: (config kwargs, batch)
    'ncsnpp_fourier_skip': (dict(), 2),
    'ncsnpp_positional_plain': (dict(nf=32, ch_mult=(1, 1, 2), attn_resolutions=(4,), embedding_type='positional',
                                     progressive='none', progressive_input='none', skip_rescale=False, centered=True), 2),
    'ncsnpp_paired_skip': (dict(name='ncsnpp_paired', channels=6, nf=32, ch_mult=(1, 2), attn_resolutions=(16,),
                                num_res_blocks=2), 2),
    # the CIFAR-10 NCSN++ configs: progressive_input = 'residual' (configs/ve/cifar10_ncsnpp_continuous.py)
    'ncsnpp_residual_input': (dict(nf=32, ch_mult=(1, 2, 2), attn_resolutions=(8,), progressive='none',
                                   progressive_input='residual'), 2),
    # the DDPM++ configs: fir = False (configs/vp/cifar10_ddpmpp_continuous.py), here with the input pyramid on (the OUTPUT pyramid
    # with fir = False cannot run in the reference: layerspp.py:117 passes 'nearest' as F.interpolate's scale_factor)
    'ncsnpp_nofir_skip': (dict(nf=32, ch_mult=(1, 2), attn_resolutions=(8,), fir=False, embedding_type='positional',
                               progressive='none'), 2),
    'ncsnpp_nofir_residual': (dict(nf=32, ch_mult=(1, 1, 2), attn_resolutions=(4,), fir=False, embedding_type='positional',
                                   progressive='none', progressive_input='residual', skip_rescale=False), 2),
}


def ncsnpp_case(case):
    kw, B = NCSNPP_CASES[case]
    cfg = make_ncsnpp_config(**kw)
    rs = np.random.RandomState(321)
    S, C = cfg.data.image_size, cfg.data.num_channels
    x = torch.from_numpy(rs.uniform(0, 1, size=(B, C, S, S)).astype(np.float32) * 3.0 - 1.0)
    if cfg.model.embedding_type == 'fourier':
        labels = torch.from_numpy(np.log(np.array([0.02, 7.5][:B], np.float32)))     # log sigma (models/utils.py:246-253)
    else:
        labels = torch.from_numpy(np.array([12.25, 871.0][:B], np.float32))          # t * (N - 1)
    return cfg, B, x, labels


def ncsnpp_params(shapes, seed, fourier_scale=16.0):
    """score_oracle.synth_params, except that a Gaussian-Fourier ``W`` keeps its real scale (the sin/cos arguments then
    reach several hundred, which is what the embedding kernel has to get right)."""
    import score_oracle as so
    p = so.synth_params(shapes, seed)
    for k in shapes:
        if k.endswith('.W') and len(shapes[k]) == 1:
            rs = np.random.RandomState(seed + 99)
            p[k] = torch.from_numpy((rs.standard_normal(shapes[k]) * fourier_scale).astype(np.float32))
    return p


def case_config(case):
    kw, B = CASES[case]
    return make_config(**kw), B


def case_y(case, B=None):
    """Synthetic condition image y in [0,1): SR-style (nearest x4 of a low-res draw) for SR3
    (mirrors lightning_data_modules/SRFLOWDataset.py:141-146), masked square for CMDE (:321-325)."""
    cfg, B0 = case_config(case)
    B = B or B0
    S = cfg.data.image_size
    rs = np.random.RandomState(123)
    if case.startswith('sr3'):
        lr = rs.uniform(0, 1, size=(B, 3, S // 4, S // 4)).astype(np.float32)
        y = np.repeat(np.repeat(lr, 4, axis=2), 4, axis=3)
    else:
        y = rs.uniform(0, 1, size=(B, 3, S, S)).astype(np.float32)
        y[:, :, S // 4:S // 4 + S // 2, S // 4:S // 4 + S // 2] = 0.
    return torch.from_numpy(y)


def tape(shapes, seed=42):
    """List of standard-normal fp32 tensors with the given shapes, in draw order."""
    rs = np.random.RandomState(seed)
    return [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in shapes]


def pc_tape_shapes(case, p_steps, B=None):
    cfg, B0 = case_config(case)
    B = B or B0
    xs = (B,) + tuple(cfg.data.shape_x)
    ys = (B,) + tuple(cfg.data.shape_y)
    shapes = [xs]
    per_phase = [ys, xs] if cfg.model.name == 'ddpm_paired' else [xs]
    for _ in range(p_steps):
        shapes += per_phase + per_phase
    return shapes


def grad_case(case):
    """Inputs of the training-loss / gradient fixtures (tests/golden/grads.npz): config with dropout off, data batch,
    fixed times and the noise tape in the loss's draw order."""
    cfg, B = case_config(case)
    cfg.model.dropout = 0.0
    rs = np.random.RandomState(11)
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    x = torch.from_numpy(rs.uniform(0, 1, size=xs).astype(np.float32))
    y = case_y(case)
    t = torch.tensor([0.83, 0.21][:B])
    shapes = [ys, xs] if cfg.model.name == 'ddpm_paired' else [xs]
    return cfg, B, x, y, t, tape(shapes, 3)


def grad_sample_index(name, numel, n=48):
    """the fixed entries of a parameter's gradient that the fixture stores"""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return np.sort(rs.choice(numel, size=min(n, numel), replace=False))


def inpaint_case():
    """inputs of the inpainting fixture: tiny unconditional net, VESDE with N = 12 steps, a half-image mask"""
    cfg, B = case_config('uncond_tiny')
    S = cfg.data.image_size
    rs = np.random.RandomState(77)
    data = torch.from_numpy(rs.uniform(0, 1, size=(B, 3, S, S)).astype(np.float32))
    mask = torch.zeros(B, 3, S, S)
    mask[:, :, :, : S // 2] = 1.
    shape = (B, 3, S, S)
    return cfg, B, data, mask, tape([shape] * (1 + 4 * 12), 23)


# ---- full-size SR3-160 (BASELINE configs[1] network), the long-schedule fixture tests/golden/sr3_160_long.npz ----
SR3_160 = dict(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), attn_resolutions=(20, 10, 5), image_size=160)
LONG_P, LONG_B, LONG_EVERY, LONG_STRIDE = 1000, 2, 50, 4


def sr3_160_y(B, seed=123):
    """synthetic LR condition: U[0,1) 20x20 -> nearest x8 (mirrors lightning_data_modules/SRFLOWDataset.py:141-146, scale 8)"""
    rs = np.random.RandomState(seed)
    lr = rs.uniform(0, 1, size=(B, 3, 20, 20)).astype(np.float32)
    return torch.from_numpy(np.repeat(np.repeat(lr, 8, axis=2), 8, axis=3))


def long_tape(P=LONG_P, B=LONG_B, seed=2024):
    """prior + 2 draws per PC step, [B,3,160,160] each, regenerated from the seed on either side"""
    return tape([(B, 3, 160, 160)] * (1 + 2 * P), seed)
