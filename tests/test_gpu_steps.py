"""The predictors / correctors beyond the fused PC pair (Euler-Maruyama, ancestral sampling, annealed Langevin
dynamics: reference sampling/predictors.py:52-76,105-179, correctors.py:111-142) on the HIP affine update
kernel, against outputs of the reference classes themselves (tests/golden/steps.npz, made by
oracle/make_goldens.py:gen_steps with a noise tape and a closed-form score)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'steps.npz')


def step_score(x, t, y=None):     # same closed form as oracle/make_goldens.py:step_score
    tt = t.reshape(-1, 1, 1, 1)
    s = -(x - 0.25) / (1.0 + tt) + 0.1 * torch.sin(3.0 * x)
    if y is not None:
        s = s + 0.05 * y
    return s


CASES = [
    ('ve_em', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'euler_maruyama', False),
    ('ve_anc', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'ancestral_sampling', False),
    ('vp_em', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'euler_maruyama', False),
    ('vp_anc', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'ancestral_sampling', False),
    ('subvp_em', 'subVPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'euler_maruyama', False),
    ('cve_em', 'cVESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'conditional_euler_maruyama', True),
    ('ve_ald', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'corrector', 'ald', False),
    ('vp_ald', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'corrector', 'ald', False),
]


class _Tape:
    def __init__(self, zs):
        self.zs, self.i = zs, 0

    def __enter__(self):
        self.orig = torch.randn_like

        def nxt(x, **k):
            z = self.zs[self.i].to(x.device)
            self.i += 1
            return z.clone()

        torch.randn_like = nxt
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


@pytest.mark.parametrize('name,scls,skw,kind,reg,cond', CASES)
def test_step_vs_reference(name, scls, skw, kind, reg, cond):
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors, predictors
    g = np.load(GOLD)
    dev = torch.device('cuda:0')
    x0, y0 = torch.from_numpy(g['x0']).to(dev), torch.from_numpy(g['y0']).to(dev)
    z0 = torch.from_numpy(g['z0'])
    sde = getattr(sde_lib, scls)(**skw)
    for ti, tv in enumerate(g['times']):
        t = torch.full((x0.shape[0],), float(tv), device=dev)
        score_fn = (lambda x, y, t: step_score(x, t, y)) if cond else (lambda x, t: step_score(x, t))
        with _Tape([z0[0], z0[1]]):
            if kind == 'predictor':
                obj = predictors.get_predictor(reg)(sde, score_fn, False)
                x, xm = obj.update_fn(x0.clone(), y0, t) if cond else obj.update_fn(x0.clone(), t)
            else:
                obj = correctors.get_corrector(reg)(sde, score_fn, 0.16, 2)
                x, xm = obj.update_fn(x0.clone(), t)
        for got, key in ((x, 'x'), (xm, 'xmean')):
            ref = torch.from_numpy(g['%s_t%d_%s' % (name, ti, key)])
            err = (got.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err < 5e-6, (name, ti, key, err)      # fp32 elementwise; coefficients rounded once on the host


def test_probability_flow_euler_is_deterministic():
    """The reference raises TypeError here (it indexes a Python float); the HIP path implements the ODE step:
    x = x_mean = x - (f - g^2/2 * score) / N, no noise."""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import predictors
    dev = torch.device('cuda:0')
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50., N=1000)
    x0 = torch.randn(2, 3, 8, 8, device=dev)
    t = torch.full((2,), 0.5, device=dev)
    obj = predictors.get_predictor('euler_maruyama')(sde, lambda x, t: step_score(x, t), True)
    x, xm = obj.update_fn(x0.clone(), t)
    g = float(sde.sde(torch.ones(1, 1, 1, 1), torch.tensor([0.5]))[1][0])
    ref = x0 + 0.5 * g * g / 1000.0 * step_score(x0, t)
    assert torch.equal(x, xm)
    assert (x - ref).abs().max().item() / ref.abs().max().item() < 5e-6


def test_generic_loop_with_other_pair_runs():
    """get_pc_sampler with a pair other than (reverse_diffusion, langevin) takes the per-step update_fn path."""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors, predictors
    dev = torch.device('cuda:0')
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=5., N=20)
    x = torch.randn(2, 3, 8, 8, device=dev) * 5
    pred = predictors.get_predictor('ancestral_sampling')(sde, lambda x, t: step_score(x, t), False)
    corr = correctors.get_corrector('ald')(sde, lambda x, t: step_score(x, t), 0.1, 1)
    for tv in np.linspace(1.0, 1e-3, 20):
        t = torch.full((2,), float(tv), device=dev)
        x, xm = corr.update_fn(x, t)
        x, xm = pred.update_fn(x, t)
    assert torch.isfinite(xm).all()


def test_global_norm_langevin_is_langevin_in_one_process():
    """Without a process group 'langevin_global' (per-sample norms on the device + host step size) must reproduce the
    fused Langevin step kernel."""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors
    dev = torch.device('cuda:0')
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50., N=1000)
    g = torch.Generator().manual_seed(11)
    x0 = (torch.randn(4, 3, 16, 16, generator=g) * 3).to(dev)
    z = torch.randn(2, 4, 3, 16, 16, generator=g)
    t = torch.full((4,), 0.4, device=dev)
    outs = []
    for name in ('langevin', 'langevin_global'):
        with _Tape([z[0], z[1]]):
            outs.append(correctors.get_corrector(name)(sde, lambda x, t: step_score(x, t), 0.16, 2).update_fn(x0.clone(), t))
    for a, b in zip(outs[0], outs[1]):
        assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item()
    n = torch.randn(5, 7, 9, device=dev)
    from conditional_score_diffusion_amd import ops
    assert torch.allclose(ops.row_norms(n), torch.norm(n.reshape(5, -1), dim=-1), rtol=1e-6)


def test_probability_flow_ode_sampler_vs_reference(golden_dir):
    """sampling.method = 'ode' (sampling/unconditional.py:93-158): scipy RK45 over the HIP drift against the reference's sample."""
    import cases
    from conditional_score_diffusion_amd.sampling.unconditional import get_sampling_fn
    from test_gpu_network import build, dev, sdes_for
    g = np.load(os.path.join(golden_dir, 'ode.npz'))
    cfg, nc, p, model = build('uncond_tiny')
    sde = sdes_for(cfg)
    cfg.sampling.method = 'ode'
    B = cases.case_config('uncond_tiny')[1]
    shape = (B,) + tuple(cfg.data.shape_x)
    z = cases.tape([shape], 17)[0] * float(cfg.model.sigma_max_x)
    sampler = get_sampling_fn(cfg, sde, shape, 1e-5)
    x, nfe = sampler(model, z=z.to(dev()))
    ref = g['x']
    assert abs(nfe - int(g['nfe'])) <= 6
    assert np.abs(x.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()


def test_pc_inpainter_vs_reference(golden_dir):
    """get_pc_inpainter (sampling/unconditional.py:230-345) on the HIP kernels against the reference's 12-step run with a noise tape:
    the known half of the image is returned exactly, the inpainted half within the trajectory tolerance."""
    import cases
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors, predictors, unconditional
    from test_gpu_network import build, dev
    g = np.load(os.path.join(golden_dir, 'inpaint.npz'))
    cfg, B, data, mask, tape = cases.inpaint_case()
    cfg, nc, p, model = build(cfg)
    sde = sde_lib.VESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 12)
    fn = unconditional.get_pc_inpainter(sde, predictors.get_predictor('reverse_diffusion'), correctors.get_corrector('langevin'),
                                        snr=0.15, n_steps=1, probability_flow=False, continuous=True, denoise=True, eps=1e-5)
    it = iter(tape)
    o_randn, o_like = torch.randn, torch.randn_like
    torch.randn = lambda *s, **k: next(it)
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        x, info = fn(model, data.to(dev()), mask.to(dev()))
    finally:
        torch.randn, torch.randn_like = o_randn, o_like
    ref = g['x']
    x = x.cpu()
    assert float(((x - data) * mask).abs().max()) == 0.0
    assert np.abs(x.numpy() - ref).max() <= 2e-4 * float(cfg.model.sigma_max_x)


# ---- Langevin corrector on the VP SDEs: step size times alphas[timestep] (sampling/correctors.py:63-67,94-98) ----
LANG_GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'vp_langevin.npz')
LANG_CASES = [      # oracle/make_goldens.py:VP_LANGEVIN_CASES (the reference's subVPSDE has no `alphas`: it raises there)
    ('vp_lang', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'langevin', False),
    ('ve_lang', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'langevin', False),
    ('cvp_lang', 'cVPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'conditional_langevin', True),
    ('cve_lang', 'cVESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'conditional_langevin', True),
]


@pytest.mark.parametrize('name,scls,skw,reg,cond', LANG_CASES)
def test_langevin_vp_and_ve_vs_reference(name, scls, skw, reg, cond):
    """two Langevin updates of the reference classes themselves (noise tape, closed-form score) at three times"""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors
    g = np.load(LANG_GOLD)
    dev = torch.device('cuda:0')
    x0, y0 = torch.from_numpy(g['x0']).to(dev), torch.from_numpy(g['y0']).to(dev)
    z0 = torch.from_numpy(g['z0'])
    sde = getattr(sde_lib, scls)(**skw)
    for ti, tv in enumerate(g['times']):
        t = torch.full((x0.shape[0],), float(tv), device=dev)
        score_fn = (lambda x, y, t: step_score(x, t, y)) if cond else (lambda x, t: step_score(x, t))
        if name + '_t%d_alpha' % ti in g.files:
            assert abs(correctors._alpha(sde, t) - float(g[name + '_t%d_alpha' % ti])) < 1e-7
        with _Tape([z0[0], z0[1]]):
            obj = correctors.get_corrector(reg)(sde, score_fn, 0.16, 2)
            x, xm = obj.update_fn(x0.clone(), y0, t) if cond else obj.update_fn(x0.clone(), t)
        for got, key in ((x, 'x'), (xm, 'xmean')):
            ref = torch.from_numpy(g['%s_t%d_%s' % (name, ti, key)])
            err = (got.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err < 5e-6, (name, ti, key, err)


def test_langevin_global_vp_equals_langevin():
    """the global-norm corrector (one process: no other ranks) carries the same alpha"""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors
    g = np.load(LANG_GOLD)
    dev = torch.device('cuda:0')
    x0 = torch.from_numpy(g['x0']).to(dev)
    z0 = torch.from_numpy(g['z0'])
    sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20., N=1000)
    t = torch.full((x0.shape[0],), 0.31, device=dev)
    outs = []
    for reg in ('langevin', 'langevin_global'):
        with _Tape([z0[0], z0[1]]):
            outs.append(correctors.get_corrector(reg)(sde, lambda x, t: step_score(x, t), 0.16, 2).update_fn(x0.clone(), t)[0])
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5 * outs[0].abs().max().item()


def test_fused_loop_corrector_alpha():
    """csd_pc_params.corr_alpha: a constant factor a on the step size equals a run with snr * sqrt(a) (step ~ snr^2 * alpha)"""
    import cases
    from test_gpu_network import build, sdes_for
    from conditional_score_diffusion_amd.sampling import fused
    case, P = 'sr3_tiny', 4
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    B = cases.case_config(case)[1]
    y = cases.case_y(case).to(torch.device('cuda:0'))
    tape = cases.tape(cases.pc_tape_shapes(case, P), seed=5)
    shape = (B,) + tuple(cfg.data.shape_x)
    a = 0.64
    xa, _, _ = fused.run(model, sde, shape, y, P, 0.16, 1e-5, True, noise_tape=tape, corr_alpha=torch.full((P,), a))
    xb, _, _ = fused.run(model, sde, shape, y, P, 0.16 * a ** 0.5, 1e-5, True, noise_tape=tape)
    x1, _, _ = fused.run(model, sde, shape, y, P, 0.16, 1e-5, True, noise_tape=tape)
    scale = xb.abs().max().item()
    assert (xa - xb).abs().max().item() < 1e-5 * scale
    assert (xa - x1).abs().max().item() > 1e-3 * scale          # (the factor does something)
