"""NCSN++ (reference models/ncsnpp.py) on the HIP per-operator path, against outputs of the reference itself
(tests/golden/ncsnpp.npz, written by oracle/make_goldens.py:gen_ncsnpp on the seeded cases of oracle/cases.py)."""
import os

import numpy as np
import pytest
import torch

import cases

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ncsnpp.npz')


@pytest.mark.parametrize('case', list(cases.NCSNPP_CASES))
def test_state_dict_layout_equals_reference(case):
    """Same keys, same order, same shapes as the reference's state_dict -> reference checkpoints load."""
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg, B, x, labels = cases.ncsnpp_case(case)
    model = mutils.create_model(cfg)
    got = ['%s|%s' % (k, ','.join(map(str, v.shape))) for k, v in model.state_dict().items()]
    want = [str(s) for s in np.load(GOLD)[case + '_keys']]
    assert got == want


def test_unsupported_options_fail_loudly():
    from conditional_score_diffusion_amd.models import utils as mutils
    with pytest.raises(NotImplementedError):
        mutils.create_model(cases.make_ncsnpp_config(progressive='residual'))
    cfg = cases.make_ncsnpp_config()
    cfg.model.progressive_combine = 'cat'
    with pytest.raises(NotImplementedError):
        mutils.create_model(cfg)
    cfg = cases.make_ncsnpp_config()
    cfg.model.resblock_type = 'ddpm'
    with pytest.raises(NotImplementedError):
        mutils.create_model(cfg)
    model = mutils.create_model(cases.make_ncsnpp_config())
    with pytest.raises(RuntimeError):                      # CPU tensors: no fallback
        model(torch.zeros(1, 3, 16, 16), torch.zeros(1))


@pytest.mark.gpu
@pytest.mark.parametrize('case', list(cases.NCSNPP_CASES))
@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('fp16x3', 2e-5), ('fp16f8', 2e-4), ('fp16', 5e-3)])
def test_forward_vs_reference(case, precision, tol):
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg, B, x, labels = cases.ncsnpp_case(case)
    cfg.model.csd_precision = precision
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(cases.ncsnpp_params(shapes, 5))
    model = model.to(dev).eval()
    x, labels = x.to(dev), labels.to(dev)
    with torch.no_grad():
        if cfg.model.name == 'ncsnpp_paired':
            r = model({'x': x[:, :3], 'y': x[:, 3:]}, labels)
            y = torch.cat([r['x'], r['y']], dim=1)
        else:
            y = model(x, labels)
    ref = torch.from_numpy(np.load(GOLD)[case + '_out'])
    err = (y.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, (case, precision, err)


@pytest.mark.gpu
def test_score_fn_and_generic_pc_sampler_run():
    """The registry, get_score_fn (Fourier label = log sigma) and the per-step PC loop accept the new family."""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg, B, x, labels = cases.ncsnpp_case('ncsnpp_fourier_skip')
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(cases.ncsnpp_params(shapes, 5))
    model = model.to(dev).eval()
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50., N=1000)
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=True)
    t = torch.tensor([0.9, 0.2], device=dev)
    s = score_fn(x.to(dev), t)
    assert s.shape == x.shape and torch.isfinite(s).all()
    # score = net(x, log sigma(t)) / sigma(t): check against a direct call
    std = sde.marginal_prob(torch.zeros(2, 1, 1, 1), t.cpu())[1].to(dev)
    with torch.no_grad():
        direct = model(x.to(dev), torch.log(std)) / std[:, None, None, None]
    assert (s - direct).abs().max().item() <= 1e-5 * direct.abs().max().item()


@pytest.mark.gpu
def test_pc_sampler_runs_on_ncsnpp():
    """sampling/unconditional.py:get_pc_sampler on the per-step path (the fused loop is DDPM-family only), with the
    reverse-diffusion + Langevin pair and with another registered pair."""
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.models import utils as mutils
    from conditional_score_diffusion_amd.sampling import correctors, predictors, unconditional
    cfg, B, x, labels = cases.ncsnpp_case('ncsnpp_fourier_skip')
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(cases.ncsnpp_params(shapes, 5))
    model = model.to(dev).eval()
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50., N=1000)
    for pred, corr in (('reverse_diffusion', 'langevin'), ('euler_maruyama', 'ald')):
        fn = unconditional.get_pc_sampler(sde, (2, 3, 16, 16), predictors.get_predictor(pred),
                                          correctors.get_corrector(corr), snr=0.075, p_steps=4, c_steps=1,
                                          continuous=True, denoise=True, eps=1e-5)
        torch.manual_seed(0)
        out, info = fn(model)
        assert out.shape == (2, 3, 16, 16) and torch.isfinite(out).all() and info['steps'] == 8


@pytest.mark.gpu
def test_larger_config_vs_oracle():
    """A configuration without a fixture (nf=64, 32x32, three levels, attention at 16, 6 -> 6 channels paired): HIP path
    vs the CPU oracle (oracle/score_oracle.py:ncsnpp_forward, itself pinned to the reference by test_oracle_golden)."""
    import score_oracle as so
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.make_ncsnpp_config(name='ncsnpp_paired', channels=6, nf=64, ch_mult=(1, 2, 2), attn_resolutions=(16,),
                                   num_res_blocks=2, image_size=32, embedding_type='positional')
    cfg.model.csd_precision = 'fp16x3'
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    p = cases.ncsnpp_params(shapes, 9)
    model.load_state_dict(p)
    model = model.to(dev).eval()
    rs = np.random.RandomState(4)
    x = torch.from_numpy(rs.uniform(-1, 2, size=(2, 6, 32, 32)).astype(np.float32))
    labels = torch.tensor([3.5, 640.0])
    with torch.no_grad():
        r = model({'x': x[:, :3].to(dev), 'y': x[:, 3:].to(dev)}, labels.to(dev))
        got = torch.cat([r['x'], r['y']], dim=1).cpu()
        ref = so.ncsnpp_forward(p, cfg, x, labels)
    assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.gpu
def test_fused_pc_loop_on_ncsnpp_vs_oracle():
    """The planned NCSN++ executor shares the fused device PC loop (csd_pc_sample) with the DDPM family: 6-step
    unconditional sampling with a noise tape and Fourier labels (log sigma) against the oracle's PC loop around
    oracle.ncsnpp_forward."""
    import score_oracle as so
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.models import utils as mutils
    from conditional_score_diffusion_amd.sampling import correctors, predictors, unconditional
    cfg, B, _, _ = cases.ncsnpp_case('ncsnpp_fourier_skip')
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    p = cases.ncsnpp_params(shapes, 5)
    model.load_state_dict(p)
    model = model.to(dev).eval()
    smin, smax, P = 0.01, 50., 6
    sde = sde_lib.VESDE(sigma_min=smin, sigma_max=smax, N=1000)
    shape = (B, 3, 16, 16)
    tp = cases.tape([shape] * (1 + 2 * P), seed=21)
    fn = unconditional.get_pc_sampler(sde, shape, predictors.get_predictor('reverse_diffusion'),
                                      correctors.get_corrector('langevin'), snr=0.075, p_steps=P, c_steps=1, continuous=True,
                                      denoise=True, eps=1e-5)
    got, info = fn(model, noise_tape=tp)             # noise_tape is only accepted by the fused path
    ve = so.VE(smin, smax, 1000)

    def score_fn(x, t):
        std = ve.std(t)
        return so.ncsnpp_forward(p, cfg, x, torch.log(std)) / std[:, None, None, None]

    with torch.no_grad():
        ref = so.pc_sample_unconditional(score_fn, shape, so.NoiseTape(tp), ve, p_steps=P, snr=0.075, eps=1e-5, denoise=True)
    assert (got.cpu() - ref).abs().max().item() / smax < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [('fp16x3', 5e-5), ('fp16f8', 3e-4), ('fp32', 5e-5), ('fp16', 3e-3)])
def test_full_size_ncsnpp_160_vs_oracle(precision, tol):
    """NCSN++ with the SR3-160 hyper-parameters (nf=96, ch_mult (1,1,2,2,3,3), attention at 20/10/5, 6 -> 6 channels,
    input/output skips, B = 2): the planned executor at the sizes the quad / loader-consumer / pointwise kernels and the
    unmasked-tile GroupNorm fusion actually run at, against the CPU oracle."""
    import score_oracle as so
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg = cases.make_ncsnpp_config(name='ncsnpp_paired', channels=6, nf=96, ch_mult=(1, 1, 2, 2, 3, 3), num_res_blocks=2,
                                   attn_resolutions=(20, 10, 5), image_size=160, embedding_type='positional')
    cfg.model.csd_precision = precision
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    p = cases.ncsnpp_params(shapes, 2)
    model.load_state_dict(p)
    model = model.to(dev).eval()
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.uniform(-1, 2, size=(2, 6, 160, 160)).astype(np.float32))
    labels = torch.tensor([17.0, 803.0])
    with torch.no_grad():
        r = model({'x': x[:, :3].to(dev), 'y': x[:, 3:].to(dev)}, labels.to(dev))
        got = torch.cat([r['x'], r['y']], dim=1).cpu()
        ref = so.ncsnpp_forward(p, cfg, x, labels)
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('executor', ['planned', 'operators'])
@pytest.mark.parametrize('case', list(cases.NCSNPP_CASES))
def test_training_mode_gradients_vs_oracle_autograd(case, executor):
    """model.train() + autograd: EVERY parameter gradient of the NCSN++ configs against torch autograd over the oracle's restatement
    (pinned to the reference's forward by tests/test_oracle_golden.py).  'planned': the whole forward / backward behind
    csd_unet_train_forward / csd_unet_backward (csrc/train_graph.h arch 1: BigGAN blocks with FIR up / down sampling and their
    transposes, Conv_2, (x + h) / sqrt(2), AttnBlockpp, Combine, input / output pyramids, Fourier / positional embedding);
    'operators': autograd over the differentiable HIP operators (FIR backward = upfirdn2d with the flipped kernel)."""
    import score_oracle as so
    from conditional_score_diffusion_amd.models import utils as mutils
    cfg, B, x, labels = cases.ncsnpp_case(case)
    cfg.model.dropout = 0.0
    dev = torch.device('cuda:0')
    model = mutils.create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    params = cases.ncsnpp_params(shapes, 5)
    model.load_state_dict(params)
    model = model.to(dev).train()
    if executor == 'planned' and not hasattr(model, '_train_forward_planned'):
        pytest.skip("progressive_input = 'residual' runs on the operator-granular class only")
    model.train_executor = executor
    w = torch.from_numpy(np.random.RandomState(2).standard_normal(tuple(x.shape[:1]) + (cfg.data.num_channels,) + tuple(x.shape[2:]))
                         .astype(np.float32))
    xd, ld = x.to(dev), labels.to(dev)
    if cfg.model.name == 'ncsnpp_paired':
        r = model({'x': xd[:, :3], 'y': xd[:, 3:]}, ld)
        out = torch.cat([r['x'], r['y']], dim=1)
    else:
        out = model(xd, ld)
    assert out.requires_grad
    # the planned executor is ONE autograd node on the library's training workspace; the operator path never allocates it
    assert (getattr(model, '_train_ws', None) is not None) == (executor == 'planned')
    (out * w.to(dev)).sum().backward()
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in params.items()}
    ref = so.ncsnpp_forward(p, cfg, x, labels)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 2e-5 * ref.abs().max().item()
    (ref * w).sum().backward()
    total = float(np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in p.values() if v.grad is not None)))
    checked = 0
    for k, q in model.named_parameters():
        if not q.requires_grad:
            assert q.grad is None                       # the Gaussian Fourier W is a fixed buffer (layerspp.py:37)
            continue
        g, r = q.grad.cpu().double(), p[k].grad.double()
        scale = max(float(r.abs().max()), float(r.norm()) / np.sqrt(r.numel()))
        assert float((g - r).abs().max()) <= 1e-3 * scale + 1e-6 * total / np.sqrt(r.numel()), (k, float((g - r).abs().max()), scale)
        checked += 1
    assert checked > 20
    model.eval()
    with torch.no_grad():
        assert torch.isfinite(model(xd, ld) if cfg.model.name != 'ncsnpp_paired' else model({'x': xd[:, :3], 'y': xd[:, 3:]}, ld)['x']).all()
