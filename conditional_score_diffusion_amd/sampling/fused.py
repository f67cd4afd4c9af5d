"""Host driver of the fused device-resident PC loop (csd_pc_sample).

Computes the per-step scalars on the CPU in fp32 with the SAME torch expressions the reference
evaluates per step (timesteps: sampling/conditional.py:202; labels: models/utils.py:213; sigma(t):
sde_lib.py:390-395; G_i: sde_lib.py:410-418), then launches ONE library call that runs the whole
loop on the device: 2 network evaluations + 2 update kernels per step, no per-step Python objects,
no host synchronisation.
"""
import ctypes

import torch

from .. import _lib, ops, sde_lib
from .._lib import check, current_stream, lib, ptr


def _rule_ids(predictor, corrector):
    """(predictor id, corrector id) of csd_pc_params, or None for classes the device loop does not implement.
    0 = the reverse-diffusion / Langevin pair, 1 = an affine rule with a per-step coefficient table, 2 = none."""
    from . import correctors as C, predictors as P
    if predictor in (P.ReverseDiffusionPredictor, P.conditionalReverseDiffusionPredictor):
        pid = 0
    elif predictor in (P.EulerMaruyamaPredictor, P.conditionalEulerMaruyamaPredictor, P.AncestralSamplingPredictor,
                       P.conditionalAncestralSamplingPredictor):
        pid = 1
    elif predictor in (P.NonePredictor, P.conditionalNonePredictor):
        pid = 2
    else:
        return None
    if corrector in (C.LangevinCorrector, C.conditionalLangevinCorrector):
        cid = 0
    elif corrector in (C.AnnealedLangevinDynamics, C.conditionalAnnealedLangevinDynamics):
        cid = 1
    elif corrector in (C.NoneCorrector, C.conditionalNoneCorrector):
        cid = 2
    else:
        return None
    return pid, cid


def fusable(model, sde, predictor, corrector, c_steps, probability_flow, continuous, use_path=False):
    """True when (model, sde, predictor, corrector) runs on the fused device loop: a VE SDE with any registered predictor
    (reverse diffusion, Euler-Maruyama, ancestral sampling, none) and corrector (Langevin, annealed Langevin dynamics, none)."""
    from ..models.ddpm import HipUNet
    from . import predictors as P
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    ok_sde = isinstance(c_sde, (sde_lib.VESDE, sde_lib.cVESDE))
    if isinstance(sde, dict):
        ok_sde = ok_sde and isinstance(sde.get('y'), sde_lib.VESDE) and len(sde) == 2
    ids = _rule_ids(predictor, corrector)
    if ids is None or ids == (2, 2):
        return False
    # the probability-flow drift exists for Euler-Maruyama only (the reverse-diffusion step kernel and ancestral sampling refuse it)
    ok_pf = (not probability_flow) or predictor in (P.EulerMaruyamaPredictor, P.conditionalEulerMaruyamaPredictor)
    # use_path (the bridge for y_t): two-SDE setting only
    ok_path = (not use_path) or isinstance(sde, dict)
    return (isinstance(model, HipUNet) and ok_sde and c_steps == 1 and ok_pf and continuous and ok_path)


def rule_tables(c_sde, ts, predictor, corrector, snr, probability_flow):
    """Per-step (p, a, b) tables of the affine rules, evaluated exactly like the per-step classes evaluate their scalars
    (sampling/predictors.py: _euler_maruyama, _ancestral; sampling/correctors.py: _ald): fp32 SDE quantities, python-float
    arithmetic, one rounding to fp32 at the library boundary."""
    from . import predictors as P
    pid, cid = _rule_ids(predictor, corrector)
    n = ts.numel()
    pred = corr = None
    if pid == 1:
        pred = torch.empty(n, 3, dtype=torch.float32)
        for i in range(n):
            t1 = ts[i:i + 1].to(torch.float32)
            if predictor in (P.EulerMaruyamaPredictor, P.conditionalEulerMaruyamaPredictor):
                drift, diffusion = c_sde.sde(torch.ones(1, 1, 1, 1), t1)
                phi, g = float(drift.flatten()[0]), float(diffusion.flatten()[0])
                dt = -1.0 / c_sde.N
                kappa = 0.5 if probability_flow else 1.0
                co = (1.0 + phi * dt, -kappa * g * g * dt, 0.0 if probability_flow else g * (-dt) ** 0.5)
            else:
                k = int((t1 * (c_sde.N - 1) / c_sde.T).long()[0])
                sig = c_sde.discrete_sigmas.to(torch.float32)
                s2 = float(sig[k]) ** 2
                a2 = float(sig[k - 1]) ** 2 if k > 0 else 0.0
                co = (1.0, s2 - a2, (a2 * (s2 - a2) / s2) ** 0.5)
            pred[i] = torch.tensor(co, dtype=torch.float64).to(torch.float32)
    if cid == 1:
        corr = torch.empty(n, 3, dtype=torch.float32)
        for i in range(n):
            t1 = ts[i:i + 1].to(torch.float32)
            std = float(c_sde.marginal_prob(torch.zeros(1, 1, 1, 1), t1)[1].flatten()[0])
            step = (snr * std) ** 2 * 2 * 1.0          # alpha = 1 for the VE SDEs
            corr[i] = torch.tensor((1.0, step, (2 * step) ** 0.5), dtype=torch.float64).to(torch.float32)
    return pid, cid, pred, corr


def step_scalars(sde, p_steps, eps, unconditional_label=None):
    """fp32 per-step arrays (labels, std_x, G, std_y|None) + the timesteps tensor."""
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    ts = torch.linspace(c_sde.T, eps, p_steps)
    dummy = torch.zeros(p_steps, 1)
    std_x = c_sde.marginal_prob(dummy, ts)[1].float()
    G = c_sde.discretize(dummy, ts)[1].float()
    if unconditional_label is None:
        labels = (ts * (c_sde.N - 1)).float()
    else:   # unconditional continuous VE: the network sees sigma(t) or log sigma(t) (models/utils.py:246-253)
        labels = torch.log(std_x) if unconditional_label == 'fourier' else std_x.clone()
    std_y = sde['y'].marginal_prob(dummy, ts)[1].float() if isinstance(sde, dict) else None
    return ts, labels.contiguous(), std_x.contiguous(), G.contiguous(), (std_y.contiguous() if std_y is not None else None)


def _fp(t):
    return t.data_ptr() and ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def fresh_seed():
    """A Philox key drawn from torch's global CPU generator: successive calls get different noise (as the reference's
    torch.randn calls do, sampling/conditional.py:198) and torch.manual_seed still makes a run reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def path_tables(sy, ts):
    """use_path: per-step bridge coefficients (w0, w1, std) of p(y_t | y_0, y_{t+tau}) and sigma_y(T + tau), evaluated in fp32 with the
    torch expressions of sde_lib.compute_backward_kernel / marginal_prob (sampling/conditional.py:143-160)"""
    tau = ts[0] - ts[1] if len(ts) > 1 else ts[0] * 0
    one, zero = torch.ones(1, 1, 1, 1), torch.zeros(1, 1, 1, 1)
    std0 = float(sy.marginal_prob(zero, (ts[0] + tau).reshape(1))[1].flatten()[0])
    rows = []
    for i in range(len(ts)):
        t1 = ts[i].reshape(1)
        m0, sd = sy.compute_backward_kernel(one, zero, t1, tau.reshape(1))
        m1, _ = sy.compute_backward_kernel(zero, one, t1, tau.reshape(1))
        rows.append([float(m0.flatten()[0]), float(m1.flatten()[0]), float(sd.flatten()[0])])
    return torch.tensor(rows, dtype=torch.float32).contiguous(), std0


def run(model, sde, shape, y, p_steps, snr, eps, denoise, noise_tape=None, seed=None, record=False,
        unconditional_label=None, global_norm=None, predictor=None, corrector=None, probability_flow=False, use_path=False,
        corr_alpha=None):
    """Run the fused loop; returns (samples, record_or_None, timesteps).  ``seed=None``: a fresh key per call (fresh_seed).

    ``predictor`` / ``corrector``: the registered classes (default: the reverse-diffusion / Langevin pair); see ``fusable``.

    ``corr_alpha``: optional [p_steps] fp32 factors of the Langevin step size (csd_pc_params.corr_alpha: alphas[timestep] of the VP
    SDEs, sampling/correctors.py:63-65,94-96); None = 1 (the VE SDEs - the only ones ``fusable`` admits today).

    ``global_norm``: None = the Langevin step size uses the batch means of THIS call's batch (the reference run on this batch;
    one library call enqueues the whole loop).  Otherwise ``(reduce_fn, global_batch)``: the batch is one shard of a larger one
    and the step size must use the means over the GLOBAL batch (identical to one reference process holding all of it,
    sampling/correctors.py:100-106): every PC step is enqueued as csd_pc_step_begin -> ``reduce_fn(sums)`` -> csd_pc_step_end,
    where ``sums`` is a 2-float device tensor and ``reduce_fn`` adds the other shards' sums into it in place (an 8-byte
    ``torch.distributed.all_reduce``) - no host synchronisation anywhere."""
    if seed is None:
        seed = fresh_seed()
    c_sde = sde['x'] if isinstance(sde, dict) else sde
    dev = model.device
    if dev.type != 'cuda':
        raise RuntimeError('the fused PC sampler runs on the MI355X only (model is on %s)' % dev)
    B = shape[0]
    ts, labels, std_x, G, std_y = step_scalars(sde, p_steps, eps, unconditional_label)
    pid = cid = 0
    pred_tab = corr_tab = None
    if predictor is not None or corrector is not None:
        from . import correctors as C_, predictors as P_
        predictor = predictor or P_.ReverseDiffusionPredictor
        corrector = corrector or C_.LangevinCorrector
        pid, cid, pred_tab, corr_tab = rule_tables(c_sde, ts, predictor, corrector, float(snr), probability_flow)
    n_phases = (pid != 2) + (cid != 2)
    path_tab, path_std0 = None, 0.0
    if use_path:
        if not isinstance(sde, dict):
            raise NotImplementedError('use_path needs the two-SDE (CMDE / VS-CMDE) setting: sde = {"x": ..., "y": ...}')
        if global_norm is not None:
            raise NotImplementedError('use_path is not provided in the global-norm sharded mode')
        path_tab, path_std0 = path_tables(sde['y'], ts)
        std_y = None                            # y_t comes from the bridge, not from the marginal
    # prior: N(0, sigma_max^2) (+ data mean) - drawn on the host like the reference (sde_lib.py:397-403)
    if noise_tape is not None:
        tape = [t.float() for t in noise_tape]
        x = (tape[0] * c_sde.sigma_max)
        if c_sde.diffused_mean is not None:
            x = x + c_sde.diffused_mean.unsqueeze(0)
        x = x.to(dev).contiguous()
        flat = torch.cat([t.reshape(-1) for t in tape[1:]]).to(dev).contiguous() if len(tape) > 1 else None
        expected = n_phases * p_steps * (2 if std_y is not None else 1)
        if use_path:                            # z_y0 | per step: z_y, z_predictor, z_corrector
            expected = 1 + p_steps * (1 + n_phases)
        if len(tape) - 1 != expected:
            raise RuntimeError('noise tape holds %d draws after the prior, the loop needs %d' % (len(tape) - 1, expected))
    else:
        x = ops.randn(tuple(shape), seed, 0, dev)
        x = ops.scale_rows(x, torch.full((B,), float(c_sde.sigma_max), device=dev))
        if c_sde.diffused_mean is not None:
            raise NotImplementedError('data-mean prior with on-device noise is not provided yet')
        flat = None
    model.eval()
    model._ensure_packed()
    ws = model._workspace(B)
    scratch = torch.empty(lib().csd_pc_scratch_bytes(model._h, B), dtype=torch.uint8, device=dev)
    rec = torch.empty((p_steps,) + tuple(x.shape), dtype=torch.float32, device=dev) if record else None
    p = _lib.PCParams()
    p.n_steps = p_steps
    p.labels, p.std_x, p.G = _fp(labels), _fp(std_x), _fp(G)
    p.std_y = _fp(std_y) if std_y is not None else None
    p.snr = float(snr)
    p.denoise = int(bool(denoise))
    p.noise_tape = flat.data_ptr() if flat is not None else None
    p.seed = int(seed)
    p.record = rec.data_ptr() if rec is not None else None
    p.predictor, p.corrector = pid, cid
    p.pred_coef = _fp(pred_tab) if pred_tab is not None else None
    p.corr_coef = _fp(corr_tab) if corr_tab is not None else None
    p.path_coef = _fp(path_tab) if path_tab is not None else None
    p.path_std0 = float(path_std0)
    if corr_alpha is not None:
        corr_alpha = torch.as_tensor(corr_alpha, dtype=torch.float32).contiguous()
        if corr_alpha.numel() != p_steps:
            raise ValueError('corr_alpha needs one factor per step (%d), got %d' % (p_steps, corr_alpha.numel()))
    p.corr_alpha = _fp(corr_alpha) if corr_alpha is not None else None
    if global_norm is not None and cid != 0:
        global_norm = None                      # only the Langevin corrector couples the samples of a batch
    yy = y.contiguous() if y is not None else None
    if global_norm is None:
        check(lib().csd_pc_sample(model._h, ptr(model._packed), ptr(ws), ws.numel(), ptr(scratch), scratch.numel(),
                                  ptr(x), ptr(yy) if yy is not None else None, B, ctypes.byref(p),
                                  current_stream(dev)), 'pc_sample')
    else:
        reduce_fn, global_batch = global_norm
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        args = (model._h, ptr(model._packed), ptr(ws), ws.numel(), ptr(scratch), scratch.numel(), ptr(x),
                ptr(yy) if yy is not None else None, B, ctypes.byref(p))
        for i in range(p_steps):
            check(lib().csd_pc_step_begin(*args, i, ptr(sums), current_stream(dev)), 'pc_step_begin')
            reduce_fn(sums)
            check(lib().csd_pc_step_end(*args, i, ptr(sums), int(global_batch), current_stream(dev)), 'pc_step_end')
    # keep the host arrays alive until the enqueue returned (they are read at enqueue time only)
    del labels, std_x, G, std_y, pred_tab, corr_tab, path_tab, corr_alpha
    return x, rec, ts
