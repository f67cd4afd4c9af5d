"""Pins the CPU oracle (oracle/score_oracle.py) to fixtures produced by the imported reference
(oracle/make_goldens.py).  Tolerances are fp32 round-off: same torch ops, possibly different
association order."""
import os

import numpy as np
import pytest
import torch

import cases
import score_oracle as so

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('case', list(cases.CASES))
def test_network_and_score(golden_dir, case):
    g = load(golden_dir, case)
    cfg, B = cases.case_config(case)
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    y = cases.case_y(case)
    ve_x = so.VE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    ve_y = so.VE(cfg.model.sigma_min_y, cfg.model.sigma_max_y, cfg.model.num_scales)
    for j, tval in enumerate([1.0, 0.5, 1e-5]):
        x = torch.from_numpy(g['x%d' % j])
        t = torch.ones(B) * tval
        with torch.no_grad():
            if cfg.model.name == 'ddpm':
                std = ve_x.std(t)
                net = so.ddpm_forward(p, nc, x, std)
                score = net / std[:, None, None, None]
            elif cfg.model.name == 'ddpm_paired':
                o = so.paired_forward(p, nc, x, y, t * 999, sr3=False)
                net = torch.cat([o['x'], o['y']], 1)
                score = so.score_paired_x(p, nc, ve_x, ve_y, x, y, t)
            else:
                net = so.paired_forward(p, nc, x, y, t * 999, sr3=True)
                score = so.score_sr3(p, nc, ve_x, x, y, t)
        assert rel(net.numpy(), g['net%d' % j]) < 2e-5
        assert rel(score.numpy(), g['score%d' % j]) < 2e-5


@pytest.mark.parametrize('case', list(cases.CASES))
@pytest.mark.parametrize('p_steps', [1, 10, 50])
def test_pc_trajectory(golden_dir, case, p_steps):
    g = load(golden_dir, case)
    cfg, B = cases.case_config(case)
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    noise = so.NoiseTape(cases.tape(cases.pc_tape_shapes(case, p_steps)))
    rec = []
    m = cfg.model
    with torch.no_grad():
        if m.name == 'ddpm':
            ve = so.VE(m.sigma_min_x, m.sigma_max_x, m.num_scales)

            def score_fn(x, t):
                std = ve.std(t)
                return so.ddpm_forward(p, nc, x, std) / std[:, None, None, None]
            xs = (B,) + tuple(cfg.data.shape_x)
            res = so.pc_sample_unconditional(score_fn, xs, noise, ve, p_steps=p_steps, snr=cfg.sampling.snr)
        else:
            res = so.pc_sample_conditional(
                p, nc, cases.case_y(case), noise, (m.sigma_min_x, m.sigma_max_x),
                (m.sigma_min_y, m.sigma_max_y) if m.name == 'ddpm_paired' else None,
                sr3=(m.name == 'ddpm_paired_SR3'), p_steps=p_steps, snr=cfg.sampling.snr,
                N=m.num_scales, record=rec)
    assert noise.i == len(noise.t)
    smax = m.sigma_max_x
    # trajectories live at scale sigma_max; compare relative to it
    err = np.abs(res.numpy() - g['pc%d' % p_steps]).max() / max(np.abs(g['pc%d' % p_steps]).max(), smax)
    assert err < 1e-4, err
    if p_steps == 10 and rec:
        ev = np.stack([r.numpy() for r in rec])
        assert np.abs(ev - g['pc10_evolution']).max() / smax < 1e-4


def test_modules(golden_dir):
    g = load(golden_dir, 'modules')
    act = torch.nn.functional.silu
    with torch.no_grad():
        for tag, cin, cout in [('res_same', 32, 32), ('res_proj', 96, 64)]:
            shapes = {'GroupNorm_0.weight': (cin,), 'GroupNorm_0.bias': (cin,),
                      'Conv_0.weight': (cout, cin, 3, 3), 'Conv_0.bias': (cout,),
                      'Dense_0.weight': (cout, 128), 'Dense_0.bias': (cout,),
                      'GroupNorm_1.weight': (cout,), 'GroupNorm_1.bias': (cout,),
                      'Conv_1.weight': (cout, cout, 3, 3), 'Conv_1.bias': (cout,)}
            if cin != cout:
                shapes.update({'NIN_0.W': (cin, cout), 'NIN_0.b': (cout,)})
            p = so.synth_params(shapes, 3)
            out = so.res_block(p, '', torch.from_numpy(g[tag + '_x']), torch.from_numpy(g[tag + '_temb']), act, cout)
            assert rel(out.numpy(), g[tag + '_out']) < 1e-5
        for tag, c in [('attn25', 64), ('attn100', 32)]:
            shapes = {'GroupNorm_0.weight': (c,), 'GroupNorm_0.bias': (c,)}
            for j in range(4):
                shapes['NIN_%d.W' % j] = (c, c)
                shapes['NIN_%d.b' % j] = (c,)
            p = so.synth_params(shapes, 4)
            out = so.attn_block(p, '', torch.from_numpy(g[tag + '_x']))
            assert rel(out.numpy(), g[tag + '_out']) < 1e-5
        shapes = {'Conv_0.weight': (32, 32, 3, 3), 'Conv_0.bias': (32,)}
        p = so.synth_params(shapes, 5)
        assert rel(so.downsample(p, '', torch.from_numpy(g['down_x']), True).numpy(), g['down_out']) < 1e-5
        assert rel(so.upsample(p, '', torch.from_numpy(g['up_x']), True).numpy(), g['up_out']) < 1e-5
        te = so.timestep_embedding(torch.from_numpy(g['temb_t']), 96)
        assert np.abs(te.numpy() - g['temb_out']).max() < 1e-6


def test_ve_scalars(golden_dir):
    g = load(golden_dir, 'sde_tables')
    ve = so.VE(5e-3, np.sqrt(np.prod([3, 160, 160])), 1000)
    assert np.array_equal(ve.discrete_sigmas.numpy(), g['discrete_sigmas'])
    for n in (50, 1000):
        ts = torch.linspace(1, 1e-5, n)
        assert np.array_equal(ts.numpy(), g['timesteps%d' % n])
        assert np.array_equal((ts * 999).numpy(), g['labels%d' % n])
        assert np.array_equal(ve.G(ts).numpy(), g['G%d' % n])
        assert np.array_equal(ve.std(ts).numpy(), g['std%d' % n])


@pytest.mark.parametrize('case', list(cases.NCSNPP_CASES))
def test_ncsnpp_oracle_vs_reference(golden_dir, case):
    """oracle.ncsnpp_forward == the reference NCSNpp.forward on the seeded cases (fixtures: oracle/make_goldens.py)."""
    g = np.load(os.path.join(golden_dir, 'ncsnpp.npz'))
    cfg, B, x, labels = cases.ncsnpp_case(case)
    shapes = {}
    for s in g[case + '_keys']:
        k, shp = str(s).split('|')
        shapes[k] = tuple(int(v) for v in shp.split(',')) if shp else ()
    p = cases.ncsnpp_params(shapes, 5)
    with torch.no_grad():
        y = so.ncsnpp_forward(p, cfg, x, labels)
    ref = torch.from_numpy(g[case + '_out'])
    assert (y - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


def oracle_loss_and_grads(case):
    """training loss + parameter gradients of the oracle (autograd over its functional restatement) on cases.grad_case"""
    cfg, B, x, y, u, tape = cases.grad_case(case)
    t = u * (1 - 1e-5) + 1e-5          # losses.py:124,190,217: t = rand * (T - eps) + eps
    nc = so.NetCfg.from_config(cfg)
    p = {k: v.clone().requires_grad_(True) for k, v in so.synth_params(so.ddpm_param_shapes(nc), 0).items()}
    ve_x = so.VE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    ve_y = so.VE(cfg.model.sigma_min_y, cfg.model.sigma_max_y, cfg.model.num_scales)
    if cfg.model.name == 'ddpm_paired':
        loss = so.dsm_loss(p, nc, cfg.model.name, ve_x, ve_y, x, y, t, tape[1], tape[0])
    else:
        loss = so.dsm_loss(p, nc, cfg.model.name, ve_x, ve_y, x, y, t, tape[0])
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in p.items()}


def check_grads_vs_fixture(g, case, loss, grads, tol):
    """loss and every parameter gradient (L2 norm + the fixture's sampled entries) against tests/golden/grads.npz"""
    assert abs(loss - float(g[case + '_loss'])) <= tol * abs(float(g[case + '_loss']))
    names = [str(n) for n in g[case + '_names']]
    assert sorted(names) == sorted(grads)
    total = float(np.sqrt((g[case + '_norms'] ** 2).sum()))
    worst = 0.0
    for i, k in enumerate(names):
        gr = grads[k].detach().reshape(-1).double().cpu().numpy()
        ref_norm = float(g[case + '_norms'][i])
        idx = cases.grad_sample_index(k, gr.size)
        # absolute floor: tensors whose gradient is tiny relative to the whole (e.g. behind init_scale=0 layers)
        floor = 1e-6 * total
        assert abs(np.sqrt((gr * gr).sum()) - ref_norm) <= tol * ref_norm + floor, (k, ref_norm)
        err = np.abs(gr[idx] - g[case + '_samples'][i][:idx.size]).max()
        scale = max(np.abs(g[case + '_samples'][i][:idx.size]).max(), ref_norm / np.sqrt(gr.size))
        assert err <= tol * scale + floor / np.sqrt(gr.size), (k, err, scale)
        worst = max(worst, err / max(scale, 1e-30))
    return worst


@pytest.mark.parametrize('case', list(cases.CASES))
def test_training_loss_and_grads(golden_dir, case):
    """a19/a20: the oracle's loss and autograd gradients reproduce the reference's (losses.py:99-232 + torch autograd)."""
    g = load(golden_dir, 'grads')
    loss, grads = oracle_loss_and_grads(case)
    check_grads_vs_fixture(g, case, loss, grads, 2e-4)


def test_probability_flow_ode_sampler(golden_dir):
    """a18 (sampling.method = 'ode'): the oracle's RK45 probability-flow sampler lands on the reference's sample with the same
    number of function evaluations."""
    g = load(golden_dir, 'ode')
    cfg, B = cases.case_config('uncond_tiny')
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    ve = so.VE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    shape = (B,) + tuple(cfg.data.shape_x)
    z = cases.tape([shape], 17)[0] * float(cfg.model.sigma_max_x)

    def score_fn(x, t):
        with torch.no_grad():
            std = ve.std(t)
            return so.ddpm_forward(p, nc, x, std) / std[:, None, None, None]

    x, nfe = so.pf_ode_sample(score_fn, ve, shape, z)
    assert abs(nfe - int(g['nfe'])) <= 6
    assert rel(x.numpy(), g['x']) < 1e-3


def test_pc_inpainter(golden_dir):
    """sampling/unconditional.py:230-345: the oracle's inpainting loop reproduces the reference's 12-step run (noise tape)."""
    g = load(golden_dir, 'inpaint')
    cfg, B, data, mask, tape = cases.inpaint_case()
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    ve = so.VE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 12)

    def score_fn(x, t):
        with torch.no_grad():
            std = ve.std(t)
            return so.ddpm_forward(p, nc, x, std) / std[:, None, None, None]

    x = so.pc_inpaint_unconditional(score_fn, data, mask, so.NoiseTape(tape), ve, snr=0.15, eps=1e-5, denoise=True)
    assert rel(x.numpy(), g['x']) < 2e-4
    assert float(((x - data) * mask).abs().max()) == 0.0
