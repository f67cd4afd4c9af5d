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


@pytest.mark.parametrize('precision,tol_n,tol_e', MODES[1:])
def test_config5_ncsnpp_256_vs_oracle(precision, tol_n, tol_e):
    """BASELINE configs[4]: NCSN++ at 256 x 256, nf = 128, ch_mult (1,1,2,2,2,2,2), attention at 16, Fourier embedding,
    input/output pyramids, 65.57 M parameters (configs/ve/ffhq_256_ncsnpp_continuous.py) - B = 2, the fp16-MFMA arithmetic modes"""
    from conditional_score_diffusion_amd.models import utils as mutils
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cfg = cases.make_ncsnpp_config(name='ncsnpp', channels=3, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                                   attn_resolutions=(16,), image_size=256, embedding_type='fourier')
    cfg.model.csd_precision = precision
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sum(int(np.prod(s)) for s in shapes.values()) == 65574549     # 65.57 M
    p = cases.ncsnpp_params(shapes, 3)
    model.load_state_dict(p)
    model = model.to(dev()).eval()
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.uniform(-1, 2, size=(2, 3, 256, 256)).astype(np.float32))
    labels = torch.tensor([np.log(3.7), np.log(0.05)], dtype=torch.float32)
    with torch.no_grad():
        got = model(x.to(dev()), labels.to(dev())).cpu()
        ref = so.ncsnpp_forward(p, cfg, x, labels)
    n, e = normwise(got.numpy(), ref.numpy()), elementwise(got.numpy(), ref.numpy())
    print('NCSN++ 256 %s: %.3e / %.3e' % (precision, n, e))
    assert n < tol_n and e < tol_e
