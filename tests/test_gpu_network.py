"""-m gpu: whole-network and sampler parity of the HIP path against (a) the committed fixtures
produced by the imported reference and (b) the CPU oracle run live on the same seeded inputs.
Tolerance (north_star): 1e-3 relative in fp32; the fp32-MFMA path is held to much tighter bounds."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
import score_oracle as so  # noqa: E402


def dev():
    return torch.device('cuda:0')


def rel(a, b, floor=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30)


def build(case_or_cfg, precision='fp32'):
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.case_config(case_or_cfg)[0] if isinstance(case_or_cfg, str) else case_or_cfg
    cfg.model.csd_precision = precision
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    model = mutils.create_model(cfg)
    missing = model.load_state_dict(p)
    assert not missing.missing_keys and not missing.unexpected_keys
    return cfg, nc, p, model.to(dev()).eval()


def sdes_for(cfg):
    from conditional_score_diffusion_amd import sde_lib
    m = cfg.model
    if m.name == 'ddpm_paired':
        return {'x': sde_lib.cVESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales),
                'y': sde_lib.VESDE(m.sigma_min_y, m.sigma_max_y, m.num_scales)}
    if m.name == 'ddpm':
        return sde_lib.VESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales)
    return sde_lib.cVESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales)


@pytest.mark.parametrize('case', list(cases.CASES))
def test_forward_and_score_vs_golden(golden_dir, case):
    from conditional_score_diffusion_amd.models import utils as mutils
    g = np.load(os.path.join(golden_dir, case + '.npz'))
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    y = cases.case_y(case).to(dev())
    B = y.shape[0]
    for j, tval in enumerate([1.0, 0.5, 1e-5]):
        x = torch.from_numpy(g['x%d' % j]).to(dev())
        t = torch.ones(B, device=dev()) * tval
        with torch.no_grad():
            if cfg.model.name == 'ddpm':
                sfn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
                score = sfn(x, t)
                net = model(x, sde.marginal_prob(x, t)[1])
            else:
                net = model({'x': x, 'y': y}, t * (cfg.model.num_scales - 1))
                sfn = mutils.get_conditional_score_fn(
                    mutils.get_score_fn(sde, model, conditional=True, train=False, continuous=True), 'x')
                score = sfn(x, y, t)
                if isinstance(net, dict):
                    net = torch.cat([net['x'], net['y']], 1)
        assert rel(net.cpu().numpy(), g['net%d' % j]) < 1e-4, (case, j)
        assert rel(score.cpu().numpy(), g['score%d' % j]) < 1e-4, (case, j)


@pytest.mark.parametrize('case', list(cases.CASES))
@pytest.mark.parametrize('p_steps', [1, 10, 50])
def test_pc_trajectory_vs_golden(golden_dir, case, p_steps):
    from conditional_score_diffusion_amd.sampling import conditional, unconditional
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    g = np.load(os.path.join(golden_dir, case + '.npz'))
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    B = cases.CASES[case][1]
    xs = (B,) + tuple(cfg.data.shape_x)
    tape = cases.tape(cases.pc_tape_shapes(case, p_steps))
    if cfg.model.name == 'ddpm':
        sampler = unconditional.get_pc_sampler(sde, xs, get_predictor('reverse_diffusion'), get_corrector('langevin'),
                                               snr=cfg.sampling.snr, p_steps=p_steps, c_steps=1, continuous=True,
                                               denoise=True, eps=1e-5)
        res, info = sampler(model, show_evolution=(p_steps == 10), noise_tape=tape)
        ev = info.get('evolution')
    else:
        sampler = conditional.get_pc_conditional_sampler(
            sde, xs, get_predictor(cfg.sampling.predictor), get_corrector(cfg.sampling.corrector),
            snr=cfg.sampling.snr, p_steps=p_steps, c_steps=1, continuous=True, denoise=True, eps=1e-5)
        res, info = sampler(model, cases.case_y(case).to(dev()), show_evolution=(p_steps == 10), noise_tape=tape)
        ev = info['evolution']['x'] if p_steps == 10 else None
    smax = cfg.model.sigma_max_x
    err = rel(res.cpu().numpy(), g['pc%d' % p_steps], floor=smax)
    assert err < 1e-3, (case, p_steps, err)          # north_star tolerance
    assert err < 2e-4, (case, p_steps, err)          # what the fp32-MFMA path should comfortably hold
    if ev is not None:
        assert np.abs(ev.numpy() - g['pc10_evolution']).max() / smax < 2e-4


@pytest.mark.parametrize('precision,tol_net,tol_traj', [('fp16x3', 1e-4, 2e-4), ('fp16f8', 3e-4, 2e-4), ('fp16', 2e-2, 1e-3)])
@pytest.mark.parametrize('case', ['sr3_tiny', 'cmde_tiny'])
def test_fp16_mfma_modes_vs_golden(golden_dir, case, precision, tol_net, tol_traj):
    """the reduced-precision conv modes against the SAME reference fixtures: split-fp16 must hold the
    fp32 bounds, plain fp16 must hold the north-star 1e-3 trajectory tolerance (50 steps)"""
    from conditional_score_diffusion_amd.sampling import conditional
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    g = np.load(os.path.join(golden_dir, case + '.npz'))
    cfg, nc, p, model = build(case, precision)
    sde = sdes_for(cfg)
    y = cases.case_y(case).to(dev())
    B = y.shape[0]
    x = torch.from_numpy(g['x1']).to(dev())
    t = torch.ones(B, device=dev()) * 0.5
    with torch.no_grad():
        net = model({'x': x, 'y': y}, t * (cfg.model.num_scales - 1))
    if isinstance(net, dict):
        net = torch.cat([net['x'], net['y']], 1)
    e_net = rel(net.cpu().numpy(), g['net1'])
    xs = (B,) + tuple(cfg.data.shape_x)
    sampler = conditional.get_pc_conditional_sampler(
        sde, xs, get_predictor(cfg.sampling.predictor), get_corrector(cfg.sampling.corrector),
        snr=cfg.sampling.snr, p_steps=50, c_steps=1, continuous=True, denoise=True, eps=1e-5)
    res, _ = sampler(model, y, noise_tape=cases.tape(cases.pc_tape_shapes(case, 50)))
    e_traj = rel(res.cpu().numpy(), g['pc50'], floor=cfg.model.sigma_max_x)
    print('precision %s %s: net err %.3e, 50-step trajectory err %.3e' % (precision, case, e_net, e_traj))
    assert e_net < tol_net and e_traj < tol_traj, (e_net, e_traj)


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('fp16x3', 2e-5), ('fp16f8', 2e-4), ('fp16', 5e-3)])
def test_nf96_batch_unmasked_tiles(precision, tol):
    """nf=96 (3 cout tiles per workgroup), 32x32 images, B=3: pixel tiles lie inside one image, so the
    fp16 kernel runs WITHOUT tap masks and must zero the halo rows that belong to the neighbouring image of
    the batch (regression: they leaked in as padding)"""
    cfg = cases.make_config(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1), num_res_blocks=1, attn_resolutions=(),
                            image_size=32)
    cfg, nc, p, model = build(cfg, precision)
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.standard_normal((3, 3, 32, 32)).astype(np.float32) * 5)
    y = torch.from_numpy(rs.uniform(0, 1, (3, 3, 32, 32)).astype(np.float32))
    lab = torch.tensor([500., 20., 900.])
    with torch.no_grad():
        ref = so.paired_forward(p, nc, x, y, lab, True)
        out = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev()))
    assert rel(out.cpu().numpy(), ref.numpy()) < tol


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-6), ('fp16', 1e-3), ('fp16x3', 2e-6), ('fp16f8', 5e-5)])
@pytest.mark.parametrize('nf,S,centered', [(96, 32, False), (128, 48, True)])
def test_fused_y_perturbation_equals_perturbing_y(precision, tol, nf, S, centered):
    """csd_unet_forward's y_noise / y_sigma (y_t = y + sigma z assembled inside the first layer: stem.hip in the fp16 modes,
    assemble_input in fp32) against the same network on a y perturbed beforehand - the first layer with and without its
    noise source, 96 and 128 output channels, both data centerings"""
    cfg = cases.make_config(name='ddpm_paired_SR3', nf=nf, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), image_size=S)
    cfg.data.centered = centered
    cfg, nc, p, model = build(cfg, precision)
    rs = np.random.RandomState(3)
    B = 3
    x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32) * 4).to(dev())
    y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32)).to(dev())
    z = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32)).to(dev())
    lab = torch.tensor([700., 30., 999.], device=dev())
    sig = 0.37
    with torch.no_grad():
        fused = model._run(x, y, lab, y_noise=z, y_sigma=sig)
        plain = model._run(x, (y + sig * z).contiguous(), lab)
        clean = model._run(x, y, lab)
    # (a fused multiply-add against torch's mul + add: one fp32 ulp of y, which the rounding of the fp16 / e4m3 operands can amplify)
    assert rel(fused.cpu().numpy(), plain.cpu().numpy()) < tol
    assert rel(clean.cpu().numpy(), plain.cpu().numpy()) > 1e-3        # the perturbation is really there


@pytest.mark.parametrize('precision,tol', [('fp16x3', 2e-5), ('fp16f8', 2e-4), ('fp16', 5e-3)])
def test_unconditional_nf96_first_layer_without_a_condition(precision, tol):
    """the unconditional family (3 input channels, y absent) through the fused first layer (stem.hip: Cy = 0), odd tile counts (48 = 3 x 16)"""
    cfg = cases.make_config(name='ddpm', nf=96, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), image_size=48)
    cfg, nc, p, model = build(cfg, precision)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((5, 3, 48, 48)).astype(np.float32) * 7)
    lab = torch.tensor([999., 500., 20., 3., 250.])
    with torch.no_grad():
        ref = so.ddpm_forward(p, nc, x, lab)
        out = model(x.to(dev()), lab.to(dev()))
    assert rel(out.cpu().numpy(), ref.numpy()) < tol


@pytest.mark.parametrize('S,B,ch_mult,attn', [(40, 3, (1, 2, 2), (20, 10)), (32, 5, (1, 1, 2, 2, 3, 3), (4, 2, 1)), (64, 2, (1, 2, 2, 3), (8,)),
                                            (80, 2, (1, 2, 3), (20,))])
def test_fp16f8_networks_vs_oracle(S, B, ch_mult, attn):
    """the fp8-correction arithmetic (CSD_PREC_F16F8) through every kernel form that carries it: conv_ff at 80 / 64 / 32 / 16 pixel
    maps, conv_f16_q with and without tap masks (tiles straddling samples at 40 / 20 / 10 / 8 / 5 / 4 / 2 / 1), the unpaired ninth
    tap, odd batches - nf = 96 networks with the SR3-160 block structure against the fp32 oracle: norm-wise 1e-4, element-wise
    4e-4 (measured ~1e-5 / 4e-5; plain fp16 sits at 7e-4 / 3e-3 on the same inputs)"""
    kw = dict(cases.SR3_160)
    kw.update(image_size=S, ch_mult=ch_mult, attn_resolutions=attn)
    cfg = cases.make_config(**kw)
    cfg, nc, p, model = build(cfg, 'fp16f8')
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32) * 20)
    y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32))
    lab = torch.full((B,), 600.)
    with torch.no_grad():
        ref = so.paired_forward(p, nc, x, y, lab, True).double()
        out = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev())).double().cpu()
    rms = float(ref.pow(2).mean().sqrt())
    nw = float((out - ref).norm() / ref.norm())
    ew = float(((out - ref).abs() / (ref.abs() + rms)).max())
    print('fp16f8 %dx%d B=%d: %.2e norm-wise, %.2e element-wise' % (S, S, B, nw, ew))
    assert nw < 1e-4 and ew < 4e-4


def _adversarial_params(p, kind, seed=11):
    """weights a trained checkpoint may have and synth_params never produces: outlier channels (x20 / x50 on ~1 % of
    the output channels of EVERY conv - they compound through the residual stream), GroupNorm gains far from 1 (x8 / x1/8 per
    channel), heavy-tailed (Student-t, 3 degrees of freedom) convolution weights"""
    rs = np.random.RandomState(seed)
    q = {}
    for k, v in p.items():
        v = v.clone()
        if kind.startswith('outlier_channels') and v.dim() == 4:
            idx = rs.choice(v.shape[0], max(1, v.shape[0] // 100), replace=False)
            v[idx] *= float(kind.rsplit('_x', 1)[1])
        elif kind == 'groupnorm_gains' and v.dim() == 1 and ('GroupNorm' in k or k.count('.') == 2) and k.endswith('weight'):
            v *= torch.from_numpy(np.where(rs.uniform(size=v.shape) < 0.5, 8.0, 0.125).astype(np.float32))
        elif kind == 'heavy_tailed' and v.dim() == 4:
            t = rs.standard_t(3, size=tuple(v.shape)).astype(np.float32)
            v = torch.from_numpy(t) * float(v.std()) / 1.7
        q[k] = v
    return q


def _adversarial_run(kind, precision):
    S, B = 80, 2
    kw = dict(cases.SR3_160)
    kw.update(image_size=S, ch_mult=(1, 2, 3), attn_resolutions=(20,))
    cfg = cases.make_config(**kw)
    cfg, nc, p, model = build(cfg, precision)
    if kind != 'large_sigma_input':
        p = _adversarial_params(p, kind)
        model.load_state_dict(p)
    rs = np.random.RandomState(3)
    scale = 3000.0 if kind == 'large_sigma_input' else 20.0
    x = torch.from_numpy(rs.standard_normal((B, 3, S, S)).astype(np.float32) * scale)
    y = torch.from_numpy(rs.uniform(0, 1, (B, 3, S, S)).astype(np.float32))
    lab = torch.full((B,), 999. if kind == 'large_sigma_input' else 600.)
    with torch.no_grad():
        ref = so.paired_forward(p, nc, x, y, lab, True).double()
        out = model({'x': x.to(dev()), 'y': y.to(dev())}, lab.to(dev())).double().cpu()
    if not torch.isfinite(out).all():
        return float('inf'), float('inf')
    rms = float(ref.pow(2).mean().sqrt())
    nw = float((out - ref).norm() / ref.norm())
    ew = float(((out - ref).abs() / (ref.abs() + rms)).max())
    print('%s %s: %.2e norm-wise, %.2e element-wise' % (precision, kind, nw, ew))
    return nw, ew


@pytest.mark.parametrize('kind', ['outlier_channels_x20', 'groupnorm_gains', 'heavy_tailed', 'large_sigma_input'])
@pytest.mark.parametrize('precision', ['fp16x3', 'fp16f8'])
def test_adversarial_weights_and_inputs_vs_oracle(kind, precision):
    """the DEFAULT arithmetic (fp16x3) holds the north-star tolerance (1e-3 norm-wise and element-wise) with a wide margin on weights /
    inputs that stress the operand ranges: outlier channels (x20 on 1 % of the couts of every conv: raw operands of a few thousand),
    extreme GroupNorm gains, heavy-tailed weights, and a network input of sigma 3000 (the resampling convs and shortcuts read the
    residual stream itself).  fp16f8 (opt-in) runs the same cases at the 1e-3 bound."""
    nw, ew = _adversarial_run(kind, precision)
    if precision == 'fp16x3':
        assert nw < 1e-4 and ew < 1e-3          # measured 1e-6 .. 2e-5 norm-wise, 4e-6 .. 2.5e-4 element-wise
    else:
        assert nw < 1e-3 and ew < 1e-3          # measured 1e-6 .. 1.3e-4 norm-wise, up to 8.6e-4 element-wise


def test_operand_range_limit_of_the_fp16_operand_modes():
    """x50 outliers on every conv drive the residual stream (the RAW operand of the resampling convs and shortcuts) past 65504, the
    largest fp16: the hi part of a split operand overflows.  Documented limit of fp16x3 / fp16f8 / fp16: such a
    network runs in csd_precision = 'fp32' (same kernels' fp32-MFMA forms, no range limit) - which must hold the tolerance here - and
    the fp16-operand modes must not return finite-but-wrong numbers silently in the default mode: fp16x3 is either accurate or
    non-finite - and the fused sampler turns non-finite into CSD_ERR_NONFINITE (test_sampler_reports_a_non_finite_state below; a
    bare network evaluation has no such check).  fp16f8's e4m3 operands saturate at +-448 instead: finite and WRONG
    (measured 1.3 norm-wise) - the reason it is not the default for unchanged reference configs."""
    nw, ew = _adversarial_run('outlier_channels_x50', 'fp32')
    assert nw < 1e-4 and ew < 1e-3
    nw3, ew3 = _adversarial_run('outlier_channels_x50', 'fp16x3')
    assert nw3 == float('inf') or (nw3 < 1e-4 and ew3 < 1e-3)


def test_sampler_reports_a_non_finite_state():
    """the finiteness contract of csd_pc_sample (include/csd.h): the x50-outlier weights overflow a raw fp16x3 operand; the fused loop
    must FAIL (CSD_ERR_NONFINITE -> NonFiniteError naming csd_precision = 'fp32') instead of returning NaN images, through the Langevin
    norms (free) as well as through the final pass over the state (a loop without a Langevin corrector); the same weights in fp32 and
    ordinary weights in fp16x3 sample normally; the step-wise global-norm form reports after its last step"""
    from conditional_score_diffusion_amd._lib import NonFiniteError
    from conditional_score_diffusion_amd.sampling import correctors as C_, fused, predictors as P_
    S, B = 80, 2
    kw = dict(cases.SR3_160)
    kw.update(image_size=S, ch_mult=(1, 2, 3), attn_resolutions=(20,))
    y = torch.from_numpy(np.random.RandomState(5).uniform(0, 1, (B, 3, S, S)).astype(np.float32)).to(dev())
    xs = (B, 3, S, S)

    def run(precision, outliers, **kwargs):
        cfg, nc, p, model = build(cases.make_config(**kw), precision)
        if outliers:
            model.load_state_dict(_adversarial_params(p, 'outlier_channels_x50'))
        sde = sdes_for(cfg)
        return fused.run(model, sde, xs, y, 2, cfg.sampling.snr, 1e-5, True, seed=11, **kwargs)[0]

    with pytest.raises(NonFiniteError, match="csd_precision = 'fp32'"):
        run('fp16x3', True)
    with pytest.raises(NonFiniteError):                 # no Langevin corrector: the final pass over the state
        run('fp16x3', True, predictor=P_.get_predictor('conditional_reverse_diffusion'), corrector=C_.get_corrector('conditional_none'))
    with pytest.raises(NonFiniteError):                 # the step-wise (global-norm) form: after its last step
        run('fp16x3', True, global_norm=(lambda sums: None, B))
    assert torch.isfinite(run('fp32', True)).all()
    assert torch.isfinite(run('fp16x3', False)).all()


def test_generic_per_step_path_matches_fused():
    """corrector/predictor objects driven step by step (reference protocol) == fused device loop"""
    from conditional_score_diffusion_amd import ops
    from conditional_score_diffusion_amd.models import utils as mutils
    from conditional_score_diffusion_amd.sampling import fused
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    case = 'sr3_tiny'
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    B = cases.CASES[case][1]
    y = cases.case_y(case).to(dev())
    tape = cases.tape(cases.pc_tape_shapes(case, 3))
    xs = (B,) + tuple(cfg.data.shape_x)
    x_f, _, _ = fused.run(model, sde, xs, y, 3, cfg.sampling.snr, 1e-5, True, noise_tape=tape)
    # same thing through update_fn objects, feeding the same noise by patching torch.randn_like
    it = iter(tape[1:])
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        sfn = mutils.get_conditional_score_fn(mutils.get_score_fn(sde, model, conditional=True, continuous=True), 'x')
        pred = get_predictor('conditional_reverse_diffusion')(sde, sfn, False)
        corr = get_corrector('conditional_langevin')(sde, sfn, cfg.sampling.snr, 1)
        x = (tape[0] * sde.sigma_max).to(dev())
        ts = torch.linspace(sde.T, 1e-5, 3)
        for i in range(3):
            vt = torch.ones(B, device=dev()) * ts[i]
            x, xm = corr.update_fn(x, y, vt)
            x, xm = pred.update_fn(x, y, vt)
    finally:
        torch.randn_like = orig
    assert rel(xm.cpu().numpy(), x_f.cpu().numpy(), floor=sde.sigma_max) < 1e-5


@pytest.mark.parametrize('pred_name,corr_name,pf', [
    ('conditional_euler_maruyama', 'conditional_langevin', False), ('conditional_ancestral_sampling', 'conditional_ald', False),
    ('conditional_none', 'conditional_ald', False), ('conditional_euler_maruyama', 'conditional_none', True),
    ('conditional_reverse_diffusion', 'conditional_none', False), ('conditional_ancestral_sampling', 'conditional_langevin', False)])
def test_other_predictors_and_correctors_on_the_fused_loop(pred_name, corr_name, pf):
    """Euler-Maruyama / ancestral sampling / annealed Langevin dynamics / none on the device-resident loop
    (csd_pc_params.predictor / .corrector + coefficient tables) == the same classes driven step by step (reference protocol,
    sampling/predictors.py:52-76,105-200, correctors.py:111-163), same noise"""
    from conditional_score_diffusion_amd.models import utils as mutils
    from conditional_score_diffusion_amd.sampling import fused
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    case, n = 'sr3_tiny', 4
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    B = cases.CASES[case][1]
    y = cases.case_y(case).to(dev())
    xs = (B,) + tuple(cfg.data.shape_x)
    P, C = get_predictor(pred_name), get_corrector(corr_name)
    assert fused.fusable(model, sde, P, C, 1, pf, True)
    phases = (pred_name != 'conditional_none') + (corr_name != 'conditional_none')
    tape = cases.tape([xs] * (1 + phases * n), 17)
    x_f, _, _ = fused.run(model, sde, xs, y, n, cfg.sampling.snr, 1e-5, True, noise_tape=tape, predictor=P, corrector=C,
                          probability_flow=pf)
    it = iter(tape[1:])
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        sfn = mutils.get_conditional_score_fn(mutils.get_score_fn(sde, model, conditional=True, continuous=True), 'x')
        pred, corr = P(sde, sfn, pf), C(sde, sfn, cfg.sampling.snr, 1)
        x = (tape[0] * sde.sigma_max).to(dev())
        ts = torch.linspace(sde.T, 1e-5, n)
        for i in range(n):
            vt = torch.ones(B, device=dev()) * ts[i]
            x, xm = corr.update_fn(x, y, vt)
            x, xm = pred.update_fn(x, y, vt)
    finally:
        torch.randn_like = orig
    assert next(it, None) is None                          # both paths consumed the same number of draws
    assert torch.isfinite(x_f).all()
    assert rel(xm.cpu().numpy(), x_f.cpu().numpy(), floor=sde.sigma_max) < 1e-5


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('fp16x3', 2e-5), ('fp16f8', 2e-4), ('fp16', 3e-3)])
def test_full_size_sr3_160_forward_vs_oracle(precision, tol):
    """cfg1/cfg2 network (nf=96, ch_mult (1,1,2,2,3,3), attention at 20/10/5) at 160x160, B=2, every precision mode
    (measured: fp32 2.1e-6, fp16x3 1.8e-6, fp16 8.6e-4):
    the shapes the quad / loader-consumer / pointwise schedules really run at"""
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cfg = cases.make_config(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1, 2, 2, 3, 3),
                            attn_resolutions=(20, 10, 5), image_size=160)
    cfg, nc, p, model = build(cfg, precision)
    rs = np.random.RandomState(5)
    lr = rs.uniform(0, 1, size=(2, 3, 20, 20)).astype(np.float32)
    y = torch.from_numpy(np.repeat(np.repeat(lr, 8, axis=2), 8, axis=3))
    for tval in (1.0, 0.3):
        sig = 5e-3 * (cfg.model.sigma_max_x / 5e-3) ** tval
        x = torch.from_numpy((rs.standard_normal((2, 3, 160, 160)) * sig + 0.5).astype(np.float32))
        labels = torch.tensor([tval * 999, tval * 412.0])
        with torch.no_grad():
            ref = so.paired_forward(p, nc, x, y, labels, sr3=True)
            out = model({'x': x.to(dev()), 'y': y.to(dev())}, labels.to(dev()))
        assert rel(out.cpu().numpy(), ref.numpy()) < tol


def test_full_size_fused_pc_steps_vs_oracle():
    """Three fused PC steps (6 network evaluations) of the 1000-step schedule at full size (SR3-160, B = 2, default
    fp16x3 arithmetic) with a noise tape against the oracle's loop: sampler kernels, batch-mean coupling and the network
    at the benchmarked shapes in one go."""
    from conditional_score_diffusion_amd.sampling import conditional
    from conditional_score_diffusion_amd.sampling.correctors import get_corrector
    from conditional_score_diffusion_amd.sampling.predictors import get_predictor
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cfg = cases.make_config(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), attn_resolutions=(20, 10, 5),
                            image_size=160)
    cfg, nc, p, model = build(cfg, 'fp16x3')
    sde = sdes_for(cfg)
    rs = np.random.RandomState(9)
    lr = rs.uniform(0, 1, size=(2, 3, 20, 20)).astype(np.float32)
    y = torch.from_numpy(np.repeat(np.repeat(lr, 8, axis=2), 8, axis=3))
    P = 3
    xs = (2, 3, 160, 160)
    tp = cases.tape([xs] * (1 + 2 * P), seed=33)
    sampler = conditional.get_pc_conditional_sampler(sde, xs, get_predictor('conditional_reverse_diffusion'),
                                                     get_corrector('conditional_langevin'), snr=cfg.sampling.snr, p_steps=P,
                                                     c_steps=1, continuous=True, denoise=True, eps=1e-5)
    got, _ = sampler(model, y.to(dev()), noise_tape=tp)
    with torch.no_grad():
        ref = so.pc_sample_conditional(p, nc, y, so.NoiseTape(tp), (cfg.model.sigma_min_x, cfg.model.sigma_max_x), None,
                                       sr3=True, p_steps=P, snr=cfg.sampling.snr, N=1000)
    assert np.abs(got.cpu().numpy() - ref.numpy()).max() / cfg.model.sigma_max_x < 2e-5


def test_batch_independence_of_network():
    """same sample, different batch position / batch size -> identical output (tiles straddle images)"""
    cfg, nc, p, model = build('sr3_tiny')
    y = cases.case_y('sr3_tiny', B=5).to(dev())
    x = torch.randn(5, 3, 20, 20, generator=torch.Generator().manual_seed(3)).to(dev()) * 30
    lab = torch.full((5,), 700.0, device=dev())
    with torch.no_grad():
        full = model({'x': x, 'y': y}, lab)
        one = model({'x': x[3:4].contiguous(), 'y': y[3:4].contiguous()}, lab[:1])
    assert torch.equal(full[3:4], one)


@pytest.mark.parametrize('case,B', [('sr3_tiny', 65), ('cmde_tiny', 64), ('uncond_tiny', 69)])
def test_batch_chunk_plan_returns_the_bits_of_the_unchunked_plan(case, B):
    """from 64 images on the planned executor runs the levels of <= 20^2 pixels as two batch chunks on two streams (csrc/unet.hip
    build_plan: OP_FORK / OP_JOIN, a private workspace block per chunk, slices of the full-batch tensors at the region's boundary):
    ragged chunk sizes (33 + 32, 35 + 34) included, every sample's output must be bit-identical to the same sample evaluated in a
    batch below the threshold (the unchunked plan), and a second call must not be disturbed by the first one's streams"""
    cfg, nc, p, model = build(case, precision='fp16x3')
    shape_x = tuple(cfg.data.shape_x) if hasattr(cfg.data, 'shape_x') else (cfg.data.num_channels, cfg.data.image_size, cfg.data.image_size)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn((B,) + shape_x, generator=g) * 20).to(dev())
    lab = (torch.rand(B, generator=g) * 900 + 50).to(dev())
    if cfg.model.name == 'ddpm':
        inp = lambda sl: x[sl].contiguous()      # noqa: E731
    else:
        y = cases.case_y(case, B=B).to(dev())
        inp = lambda sl: {'x': x[sl].contiguous(), 'y': y[sl].contiguous()}      # noqa: E731
    flat = lambda r: torch.cat([r['x'], r['y']], 1) if isinstance(r, dict) else r      # noqa: E731
    with torch.no_grad():
        full = flat(model(inp(slice(0, B)), lab))
        again = flat(model(inp(slice(0, B)), lab))
        parts = [flat(model(inp(slice(i, min(i + 8, B))), lab[i:i + 8].contiguous())) for i in range(0, B, 8)]
    assert torch.equal(full, again)
    assert torch.equal(full, torch.cat(parts))
    launches_full, launches_small = model.stats(B)[0], model.stats(8)[0]
    assert launches_full > launches_small + 20      # the region's launches appear once per chunk: the chunked plan IS what ran


def test_errors_are_loud():
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg, nc, p, model = build('sr3_tiny')
    with pytest.raises(RuntimeError):
        model({'x': torch.zeros(2, 3, 20, 20), 'y': torch.zeros(2, 3, 20, 20)}, torch.zeros(2))   # CPU tensors
    with pytest.raises(RuntimeError):
        model({'x': torch.zeros(2, 3, 16, 16, device=dev()), 'y': torch.zeros(2, 3, 16, 16, device=dev())},
              torch.zeros(2, device=dev()))
    with pytest.raises(ValueError):
        mutils.register_model(type(model), name='ddpm')


def test_use_path_sampler_vs_golden(golden_dir):
    """use_path conditional sampling (reference sampling/conditional.py:85-100,124-178: y_t follows the backward
    bridge, predictor before corrector) against the reference's own 4-step run with a noise tape
    (tests/golden/use_path.npz, oracle/make_goldens.py:gen_use_path)."""
    from conditional_score_diffusion_amd.sampling import conditional, correctors, predictors
    g = np.load(os.path.join(golden_dir, 'use_path.npz'))
    cfg, nc, p, model = build('cmde_tiny')
    sde = sdes_for(cfg)
    y = cases.case_y('cmde_tiny').to(dev())
    B = y.shape[0]
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    P = 4
    tp = cases.tape([xs, ys] + [ys, xs, xs] * P, seed=7)
    fn = conditional.get_pc_conditional_sampler(sde, xs, predictors.get_predictor('conditional_reverse_diffusion'),
                                                correctors.get_corrector('conditional_langevin'), snr=cfg.sampling.snr,
                                                p_steps=P, c_steps=1, continuous=True, denoise=True, use_path=True, eps=1e-5)
    it = iter(tp)
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *s, **k: next(it)
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        out, info = fn(model, y, show_evolution=True)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    smax = float(sde['x'].sigma_max)
    assert rel(info['evolution']['y'].numpy(), g['evo_y']) < 1e-5
    assert np.abs(info['evolution']['x'].numpy() - g['evo_x']).max() / smax < 2e-4
    assert np.abs(out.cpu().numpy() - g['out']).max() / smax < 2e-4


def test_use_path_on_the_fused_loop_vs_golden(golden_dir):
    """the same bridge sampler as ONE device-resident loop (csd_pc_params.path_coef: y_t = w0 y + w1 y_{t+tau}
    + s z once per step, predictor first, corrector on the same y_t) against the reference's own run (tests/golden/use_path.npz), and
    against the step-by-step path on the same noise"""
    from conditional_score_diffusion_amd.sampling import conditional, correctors, fused, predictors
    g = np.load(os.path.join(golden_dir, 'use_path.npz'))
    cfg, nc, p, model = build('cmde_tiny')
    sde = sdes_for(cfg)
    y = cases.case_y('cmde_tiny').to(dev())
    B = y.shape[0]
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    P = 4
    tp = cases.tape([xs, ys] + [ys, xs, xs] * P, seed=7)
    Pr, Co = predictors.get_predictor('conditional_reverse_diffusion'), correctors.get_corrector('conditional_langevin')
    assert fused.fusable(model, sde, Pr, Co, 1, False, True, use_path=True)
    fn = conditional.get_pc_conditional_sampler(sde, xs, Pr, Co, snr=cfg.sampling.snr, p_steps=P, c_steps=1, continuous=True,
                                                denoise=True, use_path=True, eps=1e-5)
    out, info = fn(model, y, noise_tape=tp)
    smax = float(sde['x'].sigma_max)
    assert info == {}
    assert np.abs(out.cpu().numpy() - g['out']).max() / smax < 2e-4
    # step by step on the same tape (the generic path draws through torch.randn / randn_like)
    it = iter(tp)
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *s, **k: next(it)
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        ref, _ = fn(model, y, show_evolution=True)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    assert np.abs(out.cpu().numpy() - ref.cpu().numpy()).max() / smax < 1e-5
    # on-device noise: reproducible per seed, different across seeds
    a, _ = fn(model, y, seed=5)
    b, _ = fn(model, y, seed=5)
    c, _ = fn(model, y, seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()


@pytest.mark.parametrize('pred_name,corr_name', [('conditional_euler_maruyama', 'conditional_none'),
                                                 ('conditional_none', 'conditional_ald'),
                                                 ('conditional_ancestral_sampling', 'conditional_langevin')])
def test_use_path_other_pairs_fused_vs_step_by_step(pred_name, corr_name):
    """the bridge mode with the other registered update rules (phases that do not exist draw nothing; the corrector is the LAST phase)"""
    from conditional_score_diffusion_amd.sampling import conditional, correctors, predictors
    cfg, nc, p, model = build('cmde_tiny')
    sde = sdes_for(cfg)
    y = cases.case_y('cmde_tiny').to(dev())
    B = y.shape[0]
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    P = 3
    phases = (pred_name != 'conditional_none') + (corr_name != 'conditional_none')
    tp = cases.tape([xs, ys] + ([ys] + [xs] * phases) * P, seed=11)
    fn = conditional.get_pc_conditional_sampler(sde, xs, predictors.get_predictor(pred_name), correctors.get_corrector(corr_name),
                                                snr=cfg.sampling.snr, p_steps=P, c_steps=1, continuous=True, denoise=True,
                                                use_path=True, eps=1e-5)
    out, _ = fn(model, y, noise_tape=tp)
    it = iter(tp)
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *s, **k: next(it)
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        ref, _ = fn(model, y, show_evolution=True)
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    assert next(it, None) is None
    assert np.abs(out.cpu().numpy() - ref.cpu().numpy()).max() / float(sde['x'].sigma_max) < 1e-5


@pytest.mark.parametrize('precision,tol', [('fp16x3', 3e-5), ('fp16f8', 3e-4), ('fp16', 5e-3), ('fp32', 3e-5)])
def test_nf128_network_vs_oracle(precision, tol):
    """nf = 128 (the VS-CMDE edges2shoes and NCSN++-256 widths): Cout = 128 / 256 run the quad schedule with four 16-cout
    tiles per N half (groups of 128 couts) instead of three; odd batch, tiles that straddle samples."""
    cfg = cases.make_config(name='ddpm_paired', nf=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16)
    cfg, nc, p, model = build(cfg, precision)
    B = 3
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.uniform(-1, 2, size=(B, 3, 16, 16)).astype(np.float32))
    y = torch.from_numpy(rs.uniform(0, 1, size=(B, 3, 16, 16)).astype(np.float32))
    labels = torch.tensor([3.0, 420.5, 998.0])
    with torch.no_grad():
        r = model({'x': x.to(dev()), 'y': y.to(dev())}, labels.to(dev()))
        got = torch.cat([r['x'], r['y']], dim=1).cpu()
        o = so.paired_forward(p, nc, x, y, labels, sr3=False)
        ref = torch.cat([o['x'], o['y']], dim=1)
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
