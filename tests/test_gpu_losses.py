"""Evaluation value of the score-matching losses (reference losses.py:55-232) on the HIP path against the reference's own
values with fixed t and noise (tests/golden/losses.npz, oracle/make_goldens.py:gen_losses)."""
import os

import numpy as np
import pytest
import torch

import cases
from test_gpu_network import build, dev, sdes_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', list(cases.CASES))
def test_loss_values_vs_reference(golden_dir, case):
    from conditional_score_diffusion_amd import losses
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    cfg, nc, p, model = build(case)
    sde = sdes_for(cfg)
    B = cases.case_config(case)[1]
    rs = np.random.RandomState(11)
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    x = torch.from_numpy(rs.uniform(0, 1, size=xs).astype(np.float32)).to(dev())
    y = cases.case_y(case).to(dev())
    tvals = torch.tensor([0.83, 0.21][:B])
    for lw in (True, False):
        if isinstance(sde, dict) and not lw:
            continue
        for rm in (True, False):
            if cfg.model.name == 'ddpm':
                fn, batch, tape = losses.get_general_sde_loss_fn(sde, False, False, rm, True, lw), x, cases.tape([xs], 3)
            elif isinstance(sde, dict):
                fn, batch, tape = losses.get_general_sde_loss_fn(sde, False, True, rm, True, lw), (y, x), cases.tape([ys, xs], 3)
            else:
                fn, batch, tape = losses.get_general_sde_loss_fn(sde, False, True, rm, True, lw), (y, x), cases.tape([xs], 3)
            it = iter(tape)
            o_rand, o_like = torch.rand, torch.randn_like
            torch.rand = lambda *a, **k: tvals.clone()
            torch.randn_like = lambda t, **k: next(it).to(t.device)
            try:
                v = float(fn(model, batch))
            finally:
                torch.rand, torch.randn_like = o_rand, o_like
            ref = float(g['%s_lw%d_rm%d' % (case, lw, rm)])
            assert abs(v - ref) <= 2e-5 * abs(ref), (case, lw, rm, v, ref)


def test_discrete_time_losses_vs_reference(golden_dir):
    """get_smld_loss_fn / get_ddpm_loss_fn (losses.py:236-265, 320-340) evaluated on the HIP path with fixed labels and noise against
    the reference's values; and get_step_fn's training branch runs (zero_grad, backward, optimize_fn, EMA update)."""
    from conditional_score_diffusion_amd import losses, optim, sde_lib
    g = np.load(os.path.join(golden_dir, 'legacy_losses.npz'))
    cfg, nc, p, model = build('uncond_tiny')
    B = cases.case_config('uncond_tiny')[1]
    rs = np.random.RandomState(11)
    xs = (B,) + tuple(cfg.data.shape_x)
    x = torch.from_numpy(rs.uniform(0, 1, size=xs).astype(np.float32)).to(dev())
    labels = torch.tensor([700, 123][:B])
    o_randint, o_like = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: labels.clone()
    try:
        for rm in (True, False):
            for lw in (True, False):
                fn = losses.get_smld_loss_fn(sde_lib.VESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales), False, rm, lw)
                it = iter(cases.tape([xs], 3))
                torch.randn_like = lambda t, **k: next(it).to(t.device)
                v = float(fn(model, x))
                ref = float(g['smld_rm%d_lw%d' % (rm, lw)])
                assert abs(v - ref) <= 2e-5 * abs(ref), ('smld', rm, lw, v, ref)
            fn = losses.get_ddpm_loss_fn(sde_lib.VPSDE(0.1, 20., cfg.model.num_scales), False, rm)
            it = iter(cases.tape([xs], 3))
            torch.randn_like = lambda t, **k: next(it).to(t.device)
            v = float(fn(model, x))
            ref = float(g['ddpm_rm%d' % rm])
            assert abs(v - ref) <= 2e-5 * abs(ref), ('ddpm', rm, v, ref)
    finally:
        torch.randint, torch.randn_like = o_randint, o_like
    # one training step through get_step_fn with the package's optimizer / EMA objects
    cfg.model.dropout = 0.0
    sde = sde_lib.VESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    state = {'model': model, 'optimizer': losses.get_optimizer(cfg, model.parameters()), 'step': 0}
    state['ema'] = optim.ExponentialMovingAverage(state['optimizer'].flat, cfg.model.ema_rate)
    step = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True, likelihood_weighting=True)
    w0 = state['optimizer'].flat.data.clone()
    l0 = float(step(state, x).detach())
    l1 = float(step(state, x).detach())
    assert np.isfinite([l0, l1]).all() and state['step'] == 2 and state['optimizer'].num_steps == 2
    assert not torch.equal(state['optimizer'].flat.data, w0)
    ev = losses.get_step_fn(sde, False, reduce_mean=True, continuous=True, likelihood_weighting=True)(state, x)
    assert np.isfinite(float(ev))
