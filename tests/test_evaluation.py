"""Evaluation harness against values computed by the reference's own lightning_callbacks/evaluation_tools.py
(tests/golden/eval_tools.npz, oracle/make_goldens.py:gen_eval), the PNG writer, and - on the GPU - a Lightning-format checkpoint
loaded, sampled from and evaluated end to end (8f rank 3 + 4)."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

import cases


def eval_case():
    rs = np.random.RandomState(2025)
    x = rs.uniform(0, 1, size=(3, 3, 40, 48)).astype(np.float32)
    x = (x + np.roll(x, 1, axis=2) + np.roll(x, 1, axis=3) + np.roll(x, 2, axis=2)) / 4.0
    s = np.clip(x + rs.standard_normal(x.shape).astype(np.float32) * 0.05, 0, 1)
    return torch.from_numpy(x), torch.from_numpy(s), np.array([[5, 7, 16], [0, 0, 20], [20, 28, 20]])


def test_metrics_match_the_reference(golden_dir):
    from conditional_score_diffusion_amd import evaluation as ev
    g = np.load(os.path.join(golden_dir, 'eval_tools.npz'))
    x, s, mask_info = eval_case()
    assert np.allclose(ev.psnr_per_image(s * 255, x * 255).numpy(), g['psnr_each'], rtol=1e-6)
    assert ev.mean_psnr(s * 255, x * 255) == pytest.approx(float(g['mean_psnr']), rel=1e-6)
    assert np.allclose(ev.ssim_per_image(s * 255, x * 255).numpy(), g['ssim_each'], rtol=1e-6)
    assert ev.mean_ssim(s * 255, x * 255) == pytest.approx(float(g['mean_ssim']), rel=1e-6)
    for scale in (0.25, 0.125, 2.0):
        ref = g['resize_%g' % scale]
        got = ev.resize(x, scale).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-6, scale
    assert ev.resize(x[0], 0.25).shape == (3, 10, 12)
    assert ev.get_calculate_consistency_fn('super-resolution')(s, x, 4) == pytest.approx(float(g['consistency_sr']), rel=1e-5)
    assert ev.get_calculate_consistency_fn('inpainting')(s, x, mask_info) == pytest.approx(float(g['consistency_inp']), rel=1e-6)
    assert ev.psnr_per_image(x, x)[0] == float('inf')
    with pytest.raises(NotImplementedError):
        ev.get_calculate_consistency_fn('image-to-image')


def test_ssim_against_an_independent_scipy_restatement():
    """SSIM pin.  The reference's calculate_ssim needs OpenCV, which this image lacks: the fixture value in eval_tools.npz came from
    the reference's function running on a builder-written cv2 stand-in (oracle/make_goldens.py) - i.e. NOT pinned to OpenCV itself.
    Second opinion here: the published formula (Wang et al. 2004: 11 x 11 Gaussian window, sigma 1.5, K1 = 0.01, K2 = 0.03, L = 255,
    'valid' region, mean over the map, mean over the channels) restated with scipy's 2-D convolution, independent of the module's
    torch code.  SSIM therefore stays 'unpinned against OpenCV', cross-checked against scipy."""
    from scipy.signal import convolve2d
    from conditional_score_diffusion_amd import evaluation as ev
    x, s, _ = eval_case()
    a, b = (s * 255).double().numpy(), (x * 255).double().numpy()
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    win = np.outer(g, g)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    want = []
    for i in range(a.shape[0]):
        per_c = []
        for c in range(a.shape[1]):
            f = lambda t: convolve2d(t, win, mode='valid')      # noqa: E731
            m1, m2 = f(a[i, c]), f(b[i, c])
            s1, s2, s12 = f(a[i, c] ** 2) - m1 * m1, f(b[i, c] ** 2) - m2 * m2, f(a[i, c] * b[i, c]) - m1 * m2
            per_c.append((((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2))).mean())
        want.append(np.mean(per_c))
    assert np.allclose(ev.ssim_per_image(s * 255, x * 255).numpy(), want, rtol=1e-9)
    assert ev.ssim_per_image(x * 255, x * 255).numpy() == pytest.approx(1.0, abs=1e-12)


@pytest.mark.gpu
def test_metrics_match_the_reference_on_device_tensors(golden_dir):
    """the same reference values with the images ON THE GPU (what the evaluator sees after sample()): the metrics are torch code
    (F.conv2d / matmul in float64 on the samples' device - plumbing around the HIP path, not part of it)"""
    from conditional_score_diffusion_amd import evaluation as ev
    g = np.load(os.path.join(golden_dir, 'eval_tools.npz'))
    d = torch.device('cuda:0')
    x, s, mask_info = eval_case()
    x, s = x.to(d), s.to(d)
    assert np.allclose(ev.psnr_per_image(s * 255, x * 255).cpu().numpy(), g['psnr_each'], rtol=1e-5)
    assert np.allclose(ev.ssim_per_image(s * 255, x * 255).cpu().numpy(), g['ssim_each'], rtol=1e-5)
    for scale in (0.25, 0.125, 2.0):
        ref = g['resize_%g' % scale]
        got = ev.resize(x, scale).cpu().numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 5e-6, scale
    assert ev.get_calculate_consistency_fn('super-resolution')(s, x, 4) == pytest.approx(float(g['consistency_sr']), rel=1e-4)
    assert ev.get_calculate_consistency_fn('inpainting')(s, x, mask_info) == pytest.approx(float(g['consistency_inp']), rel=1e-5)


def test_png_writer_roundtrip(tmp_path):
    from conditional_score_diffusion_amd import evaluation as ev
    t = torch.rand(3, 7, 5, generator=torch.Generator().manual_seed(0))
    p = os.path.join(tmp_path, 'a', 'img.png')
    ev.save_image(t, p)
    raw = open(p, 'rb').read()
    assert raw[:8] == b'\x89PNG\r\n\x1a\n'
    W, H, depth, ctype = struct.unpack('>IIBB', raw[16:26])
    assert (W, H, depth, ctype) == (5, 7, 8, 2)
    i = raw.index(b'IDAT')
    n = struct.unpack('>I', raw[i - 4:i])[0]
    px = np.frombuffer(zlib.decompress(raw[i + 4:i + 4 + n]), np.uint8).reshape(7, 1 + 15)[:, 1:].reshape(7, 5, 3)
    assert np.array_equal(px, (t.permute(1, 2, 0) * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy())


class _FakeModule:
    """sample() = ground truth + draw-dependent noise: drives the draws x snr bookkeeping without a GPU"""
    def __init__(self, x):
        self.x, self.calls = x, []
        self.config = cases.make_config()
        self.config.data.scale = 4

    def sample(self, y, show_evolution=False, snr='default', **kw):
        self.calls.append(snr)
        g = torch.Generator().manual_seed(len(self.calls))
        return self.x + torch.randn(self.x.shape, generator=g) * float(snr), {}


def test_draws_times_snr_loop(tmp_path):
    """PairedCallback.py:158-232: every snr x every draw samples once; clamp; per-batch means per snr; diversity needs > 1 draw"""
    from conditional_score_diffusion_amd import evaluation as ev
    x, s, _ = eval_case()
    mod = _FakeModule(x)
    e = ev.PairedEvaluator(snr=(0.05, 0.2), draws=3, task='super-resolution', save_samples_dir=os.path.join(tmp_path, 'samples'))
    e.evaluate_batch(mod, torch.zeros_like(x), x)
    e.evaluate_batch(mod, torch.zeros_like(x), x)
    assert mod.calls == [0.05] * 3 + [0.2] * 3 + [0.05] * 3 + [0.2] * 3 and e.images_tested == 6
    summ = e.summary()
    assert summ[0.05]['psnr'] > summ[0.2]['psnr'] and summ[0.05]['ssim'] > summ[0.2]['ssim']      # less noise, better scores
    assert summ[0.2]['diversity'] > summ[0.05]['diversity'] > 0 and summ[0.05]['consistency'] > summ[0.2]['consistency']
    assert os.path.exists(os.path.join(tmp_path, 'samples', 'snr_0.200', 'draw_3', '6.png'))
    one = ev.PairedEvaluator(snr=(0.1,), draws=1)
    assert 'diversity' not in one.results[0.1]
    with pytest.raises(NotImplementedError):
        ev.PairedEvaluator(evaluation_metrics=('lpips',))


@pytest.mark.gpu
def test_checkpoint_to_samples_to_metrics_on_gpu(tmp_path):
    """a Lightning-format .ckpt (score_model.* weights, hyper_parameters.config, the VS-CMDE sigma_max_y / sigma_min_y buffers) ->
    load_score_module -> sde['y'] from the buffers -> sample() on the GPU == the same run with hand-built objects and == the
    oracle's loop with those sigmas; then the evaluator runs draws x snr on it"""
    import score_oracle as so
    from conditional_score_diffusion_amd import checkpoint, evaluation as ev, sde_lib
    from conditional_score_diffusion_amd.models import utils as mutils
    dev = torch.device('cuda:0')
    cfg, B = cases.case_config('cmde_tiny')
    cfg.data.use_data_mean = False
    cfg.model.num_scales = 1000
    nc = so.NetCfg.from_config(cfg)
    p = so.synth_params(so.ddpm_param_shapes(nc), 0)
    sd = {'score_model.' + k: v for k, v in p.items()}
    sd['sigma_max_y'], sd['sigma_min_y'] = torch.tensor(0.37), torch.tensor(0.004)
    path = os.path.join(tmp_path, 'epoch=7-step=1234.ckpt')
    torch.save({'state_dict': sd, 'hyper_parameters': {'config': cfg}, 'epoch': 7}, path)
    mod = checkpoint.load_score_module(path, device=dev)
    assert mod.sde['y'].sigma_max == pytest.approx(0.37) and next(mod.score_model.parameters()).is_cuda
    y = cases.case_y('cmde_tiny').to(dev)
    P = 6
    tape = cases.tape(cases.pc_tape_shapes('cmde_tiny', P), seed=5)
    got, _ = mod.sample(y, p_steps=P, noise_tape=tape)
    with torch.no_grad():
        ref = so.pc_sample_conditional(p, nc, y.cpu(), so.NoiseTape(tape), (cfg.model.sigma_min_x, cfg.model.sigma_max_x), (0.004, 0.37),
                                       sr3=False, p_steps=P, snr=cfg.sampling.snr, N=1000)
    assert (got.cpu() - ref).abs().max().item() / cfg.model.sigma_max_x < 2e-4
    # a DIFFERENT sigma_max_y gives a different sample: the buffers really are what configured the SDE
    other = checkpoint.ScoreModule(cfg, mod.score_model, {'x': mod.sde['x'], 'y': sde_lib.VESDE(0.004, 1.0, 1000)}, 1e-5)
    o2, _ = other.sample(y, p_steps=P, noise_tape=tape)
    assert (o2 - got).abs().max().item() / cfg.model.sigma_max_x > 1e-3
    e = ev.PairedEvaluator(evaluation_metrics=('psnr', 'ssim', 'consistency', 'diversity'), snr=(0.1, 0.2), draws=2, task='inpainting',
                           p_steps=4, sampler_kw={'seed': 3})
    x = torch.rand(B, 3, 20, 20, device=dev)
    res = e.evaluate_batch(mod, y, x, mask_info=np.array([[5, 5, 10]] * B))
    assert all(np.isfinite(v).all() for r in res.values() for v in r.values()) and e.images_tested == B
