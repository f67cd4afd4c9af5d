"""Test-time evaluation around ``sample()``: the draws x snr loop of the reference's test callback and its
image metrics, computed on the device the samples already live on.

Mirrors (behaviour only) lightning_callbacks/PairedCallback.py:158-232 - for every snr and every draw: sample x | y, clamp to
[0, 1], optionally write PNGs, then PSNR / SSIM / consistency / diversity on [0, 255] images - and
lightning_callbacks/evaluation_tools.py: ``calculate_psnr`` / ``calculate_mean_psnr`` (:67-84), ``ssim`` / ``calculate_ssim`` /
``calculate_mean_ssim`` (:93-143: 11x11 Gaussian window, sigma 1.5, 'valid' region, per channel, the constants of Wang et al.),
``resize`` / ``imresize`` (:177-318: MATLAB-style bicubic with antialiasing, symmetric borders) behind the super-resolution
consistency (:14-20) and the masked PSNR of the inpainting consistency (:21-34).  LPIPS needs a pretrained AlexNet (no network in the
image) and the Canny-edge consistency of image-to-image translation needs OpenCV: both are outside this module.

Everything is torch on the samples' device in float64 (the reference converts to float64 numpy on the host); the bicubic resize
is two dense matrix products per image batch (the reference loops over rows and channels in Python).
"""
import math
import os
import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------------------------
# metrics on [0, 255] images, NCHW (the reference feeds NHWC-swapped numpy arrays; the values do not depend on the layout)
# ------------------------------------------------------------------------------------------------------------------------------
def psnr_per_image(a, b):
    """evaluation_tools.calculate_psnr (:67-74) for every image of a batch: 20 log10(255 / sqrt(mse)), inf for identical images"""
    a, b = a.double(), b.double()
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return torch.where(mse == 0, torch.full_like(mse, float('inf')), 20.0 * torch.log10(255.0 / torch.sqrt(mse)))


def mean_psnr(a, b):
    """evaluation_tools.calculate_mean_psnr (:76-84)"""
    return float(psnr_per_image(a, b).mean())


def _gaussian_window(device):
    g = torch.arange(11, dtype=torch.float64, device=device) - 5.0
    g = torch.exp(-(g * g) / (2.0 * 1.5 * 1.5))         # cv2.getGaussianKernel(11, 1.5)
    g = g / g.sum()
    return torch.outer(g, g)


def ssim_per_image(a, b):
    """evaluation_tools.calculate_ssim (:118-135): the single-scale SSIM index of Wang et al. per channel (Gaussian 11 x 11 window,
    'valid' part), averaged over the channels.  a, b: [B, C, H, W] in [0, 255], H, W >= 11."""
    a, b = a.double(), b.double()
    B, C, H, W = a.shape
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    w = _gaussian_window(a.device).reshape(1, 1, 11, 11)
    flt = lambda t: F.conv2d(t.reshape(B * C, 1, H, W), w)        # noqa: E731  (symmetric window: correlation = convolution)
    mu1, mu2 = flt(a), flt(b)
    s1, s2, s12 = flt(a * a) - mu1 * mu1, flt(b * b) - mu2 * mu2, flt(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.flatten(1).mean(1).reshape(B, C).mean(1)


def mean_ssim(a, b):
    """evaluation_tools.calculate_mean_ssim (:137-143)"""
    return float(ssim_per_image(a, b).mean())


# ------------------------------------------------------------------------------------------------------------------------------
# MATLAB-style bicubic resize with antialiasing (evaluation_tools.resize / imresize / calculate_weights_indices, :177-318)
# ------------------------------------------------------------------------------------------------------------------------------
def _cubic(x):
    ax = x.abs()
    ax2, ax3 = ax * ax, ax * ax * ax
    return (1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1).to(x.dtype) + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) & (ax <= 2)).to(x.dtype)


def resize_matrix(in_len, scale, antialiasing=True, dtype=torch.float32):
    """[out_len, in_len] matrix of the 1-D resize: row k holds the normalised cubic weights of output pixel k, with the taps that
    fall off either end folded back onto the image (symmetric border)"""
    out_len = math.ceil(in_len * scale)
    kw = 4.0 / scale if (scale < 1 and antialiasing) else 4.0
    x = torch.linspace(1, out_len, out_len, dtype=dtype)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = torch.floor(u - kw / 2)
    P = math.ceil(kw) + 2
    idx = left.view(-1, 1) + torch.linspace(0, P - 1, P, dtype=dtype).view(1, -1)           # 1-based input coordinates
    d = u.view(-1, 1) - idx
    wts = scale * _cubic(d * scale) if (scale < 1 and antialiasing) else _cubic(d)
    wts = wts / wts.sum(1, keepdim=True)
    idx = idx.long() - 1                                                                     # 0-based, may leave [0, in_len)
    period = 2 * in_len                                                                      # symmetric (mirror incl. the edge pixel)
    idx = idx % period
    idx = torch.where(idx >= in_len, period - 1 - idx, idx)
    M = torch.zeros(out_len, in_len, dtype=dtype)
    M.scatter_add_(1, idx, wts)
    return M


def resize(img, scale, antialiasing=True):
    """evaluation_tools.resize (:177-187): [C, H, W] or [B, C, H, W] in, same rank out, H and W scaled by ``scale``"""
    squeeze = img.dim() == 3
    if img.dim() not in (3, 4):
        raise NotImplementedError('img dimension not supported.')
    x = img.unsqueeze(0) if squeeze else img
    MH = resize_matrix(x.shape[2], scale, antialiasing).to(x.device, x.dtype)
    MW = resize_matrix(x.shape[3], scale, antialiasing).to(x.device, x.dtype)
    out = torch.matmul(torch.matmul(MH, x), MW.t())
    return out.squeeze(0) if squeeze else out


def get_calculate_consistency_fn(task):
    """evaluation_tools.get_calculate_consistency_fn (:14-66): apply the forward operator to the samples and to the ground truth and
    compare - PSNR of the bicubic-downscaled images (super-resolution), PSNR outside the mask (inpainting)"""
    if task == 'super-resolution':
        def consistency_fn(samples, hr_gt, scale):
            return mean_psnr(resize(samples, 1 / scale) * 255, resize(hr_gt, 1 / scale) * 255)
    elif task == 'inpainting':
        def consistency_fn(samples, gt, mask_info):
            s, g = samples.clone(), gt.clone()
            for i in range(s.size(0)):
                sx, sy, ms = int(mask_info[i, 0]), int(mask_info[i, 1]), int(mask_info[i, 2])
                s[i, :, sx:sx + ms, sy:sy + ms] = 0.
                g[i, :, sx:sx + ms, sy:sy + ms] = 0.
            return mean_psnr(s * 255, g * 255)
    else:
        raise NotImplementedError('The forward operator for task %s is not supported.' % task)
    return consistency_fn


# ------------------------------------------------------------------------------------------------------------------------------
# PNG writer (torchvision.utils.save_image of one [3|1, H, W] tensor in [0, 1]; evaluation_tools.py:145-176 writes through OpenCV)
# ------------------------------------------------------------------------------------------------------------------------------
def save_image(tensor, fp):
    t = tensor.detach().float().cpu()
    if t.dim() == 2:
        t = t.unsqueeze(0)
    arr = t.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    H, W, C = arr.shape
    if C not in (1, 3):
        raise ValueError('save_image: 1 or 3 channels, got %d' % C)
    raw = b''.join(b'\x00' + arr[r].tobytes() for r in range(H))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    os.makedirs(os.path.dirname(os.path.abspath(fp)), exist_ok=True)
    with open(fp, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2 if C == 3 else 0, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


# ------------------------------------------------------------------------------------------------------------------------------
# the draws x snr loop
# ------------------------------------------------------------------------------------------------------------------------------
class PairedEvaluator:
    """TestPairedVisualizationCallback (PairedCallback.py:120-232) without Lightning: ``evaluate_batch(module, y, x)`` runs, for every
    snr of ``snr`` and every draw, ``module.sample(y, ...)`` (a checkpoint.ScoreModule or anything with that method), clamps to
    [0, 1] and accumulates the metrics; ``summary()`` gives the per-snr means the reference prints at the end of the test epoch."""

    def __init__(self, evaluation_metrics=('psnr', 'ssim', 'consistency', 'diversity'), snr=(0.15,), draws=1, task='super-resolution',
                 scale=None, predictor='default', corrector='default', p_steps='default', c_steps='default', denoise='default',
                 use_path='default', save_samples_dir=None, sampler_kw=None):
        self.metrics = [m.lower() for m in evaluation_metrics]
        for m in self.metrics:
            if m not in ('psnr', 'ssim', 'consistency', 'diversity'):
                raise NotImplementedError('metric %r is not provided (lpips needs pretrained AlexNet weights)' % m)
        self.snr = list(snr)
        self.draws = list(range(1, draws + 1)) if isinstance(draws, int) else list(draws)
        self.task, self.scale = task, scale
        self.sample_kw = dict(predictor=predictor, corrector=corrector, p_steps=p_steps, c_steps=c_steps, denoise=denoise,
                              use_path=use_path)
        self.sampler_kw = dict(sampler_kw or {})
        self.dir = save_samples_dir
        self.results = {s: {m: [] for m in self.metrics if not (m == 'diversity' and len(self.draws) == 1)} for s in self.snr}
        self.images_tested = 0

    def generate_metric_vals(self, y, x, module, snr, mask_info=None):
        vals = {m: [] for m in self.metrics if not (m == 'diversity' and len(self.draws) == 1)}
        cons = get_calculate_consistency_fn(self.task) if 'consistency' in self.metrics else None
        for draw in self.draws:
            samples, _ = module.sample(y, show_evolution=False, snr=snr, **self.sample_kw, **self.sampler_kw)
            samples = torch.clamp(samples, min=0, max=1)      # (the model is p_epsilon, not p_0: values slightly off are corrected)
            if self.dir:
                for i in range(samples.size(0)):
                    save_image(samples[i], os.path.join(self.dir, 'snr_%.3f' % snr, 'draw_%d' % draw, '%d.png' % (self.images_tested + i + 1)))
            if 'psnr' in vals:
                vals['psnr'] = mean_psnr(samples * 255, x * 255)          # (the reference keeps the LAST draw's value here)
            if 'ssim' in vals:
                vals['ssim'].append(mean_ssim(samples * 255, x * 255))
            if 'consistency' in vals:
                if self.task == 'super-resolution':
                    scale = self.scale if self.scale is not None else module.config.data.scale
                    vals['consistency'].append(cons(samples, x, scale))
                else:
                    vals['consistency'].append(cons(samples, x, mask_info))
            if 'diversity' in vals:
                vals['diversity'].append(samples * 255.)
        return vals

    def evaluate_batch(self, module, y, x, mask_info=None):
        for e_snr in self.snr:
            vals = self.generate_metric_vals(y, x, module, e_snr, mask_info)
            for m, v in vals.items():
                if m == 'diversity':
                    self.results[e_snr][m].append(float(torch.mean(torch.std(torch.stack(v), dim=0))))
                else:
                    self.results[e_snr][m].append(float(np.mean(v)))
        self.images_tested += x.size(0)
        return self.results

    def summary(self):
        return {s: {m: float(np.mean(v)) for m, v in r.items() if v} for s, r in self.results.items()}
