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
