"""Training side of the path: the backward kernels of csrc/backward.hip behind the differentiable
operators of grad_ops, and the training loss + parameter gradients of the whole network against the reference's own autograd
(tests/golden/grads.npz, oracle/make_goldens.py:gen_grads) and against the oracle's autograd.  Tolerance: 1e-3 relative fp32
(north star); the per-operator checks are tighter."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
import score_oracle as so
from test_gpu_network import build, dev, sdes_for
from test_oracle_golden import check_grads_vs_fixture, oracle_loss_and_grads

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(rs, *shape):
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


CONV_CASES = [  # B, Cin, Cout, H, k, stride, up2
    (2, 32, 32, 16, 3, 1, False), (3, 6, 32, 12, 3, 1, False), (2, 32, 3, 12, 3, 1, False), (2, 40, 72, 10, 3, 1, False),
    (2, 64, 32, 8, 1, 1, False), (2, 32, 32, 16, 3, 2, False), (2, 32, 64, 8, 3, 1, True), (1, 96, 96, 5, 3, 1, False),
]


@pytest.mark.parametrize('B,Cin,Cout,H,k,stride,up2', CONV_CASES)
def test_conv2d_backward_vs_torch(B, Cin, Cout, H, k, stride, up2):
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(5)
    x, w, b = rnd(rs, B, Cin, H, H), rnd(rs, Cout, Cin, k, k) * 0.1, rnd(rs, Cout)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    u = F.interpolate(xr, scale_factor=2, mode='nearest') if up2 else xr
    if stride == 2:
        ref = F.conv2d(F.pad(u, (0, 1, 0, 1)), wr, br, stride=2)
    else:
        ref = F.conv2d(u, wr, br, padding=k // 2)
    dy = rnd(rs, *ref.shape)
    ref.backward(dy)
    xd, wd, bd = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    out = G.conv2d(xd, wd, bd, stride=stride, downsample_pad=stride == 2, up2=up2)
    assert rel(out, ref) < 1e-5
    out.backward(dy.to(dev()))
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(wd.grad, wr.grad) < 1e-5
    assert rel(bd.grad, br.grad) < 1e-5


def test_conv2d_wgrad_is_deterministic_and_large():
    """split-K partials are reduced in a fixed order: two runs agree bit for bit; a level-0-sized layer against torch"""
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(6)
    x, w = rnd(rs, 4, 96, 64, 64), rnd(rs, 96, 96, 3, 3) * 0.05
    dy = rnd(rs, 4, 96, 64, 64)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, padding=1).backward(dy)
    got = []
    for _ in range(2):
        wd = w.to(dev()).requires_grad_(True)
        G.conv2d(x.to(dev()), wd, None).backward(dy.to(dev()))
        got.append(wd.grad.clone())
    assert torch.equal(got[0], got[1])
    assert rel(got[0], wr.grad) < 2e-5


@pytest.mark.parametrize('B,C,H,groups,act', [(2, 32, 8, 32, 'swish'), (3, 64, 5, 32, 'none'), (2, 96, 12, 32, 'swish'),
                                               (2, 48, 6, 12, 'swish')])
def test_groupnorm_act_backward_vs_torch(B, C, H, groups, act):
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(7)
    x, ga, be = rnd(rs, B, C, H, H) * 2 + 0.3, rnd(rs, C) * 0.5 + 1, rnd(rs, C) * 0.2
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, ga, be))
    ref = F.group_norm(xr, groups, gr, br, eps=1e-6)
    ref = F.silu(ref) if act == 'swish' else ref
    dy = rnd(rs, *ref.shape)
    ref.backward(dy)
    xd, gd, bd = (t.to(dev()).requires_grad_(True) for t in (x, ga, be))
    out = G.groupnorm_act(xd, gd, bd, groups, 1e-6, act)
    assert rel(out, ref) < 1e-5
    out.backward(dy.to(dev()))
    assert rel(xd.grad, xr.grad) < 2e-5
    assert rel(gd.grad, gr.grad) < 2e-5
    assert rel(bd.grad, br.grad) < 2e-5


@pytest.mark.parametrize('B,C,H', [(2, 32, 4), (2, 64, 8), (1, 96, 10), (2, 288, 5)])
def test_attention_backward_vs_torch(B, C, H):
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(8)
    q, k, v = rnd(rs, B, C, H, H), rnd(rs, B, C, H, H), rnd(rs, B, C, H, H)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    w = torch.einsum('bchw,bcij->bhwij', qr, kr) * (int(C) ** (-0.5))        # models/layers.py:584-588
    w = F.softmax(w.reshape(B, H, H, H * H), dim=-1).reshape(B, H, H, H, H)
    ref = torch.einsum('bhwij,bcij->bchw', w, vr)
    do = rnd(rs, *ref.shape)
    ref.backward(do)
    qd, kd, vd = (t.to(dev()).requires_grad_(True) for t in (q, k, v))
    out = G.attention(qd, kd, vd)
    assert rel(out, ref) < 1e-5
    out.backward(do.to(dev()))
    assert rel(qd.grad, qr.grad) < 2e-5
    assert rel(kd.grad, kr.grad) < 2e-5
    assert rel(vd.grad, vr.grad) < 2e-5


@pytest.mark.parametrize('B,K,N,act', [(2, 32, 128, 'none'), (5, 128, 128, 'swish'), (3, 512, 96, 'swish')])
def test_linear_backward_vs_torch(B, K, N, act):
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(9)
    x, w, b = rnd(rs, B, K), rnd(rs, N, K) * 0.1, rnd(rs, N)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(F.silu(xr) if act == 'swish' else xr, wr, br)
    dy = rnd(rs, B, N)
    ref.backward(dy)
    xd, wd, bd = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    out = G.linear(xd, wd, bd, act_in=act)
    out.backward(dy.to(dev()))
    assert rel(out, ref) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(wd.grad, wr.grad) < 1e-5
    assert rel(bd.grad, br.grad) < 1e-5


def test_small_differentiable_ops():
    from conditional_score_diffusion_amd import grad_ops as G
    rs = np.random.RandomState(10)
    x, bias, z = rnd(rs, 3, 8, 5, 5), rnd(rs, 3, 8), rnd(rs, 3, 8, 5, 5)
    sc = torch.tensor([0.5, 2.0, 3.0])
    xr, br = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = (((xr + br[:, :, None, None]) * 2 - 1) / np.sqrt(2.) + z) / sc[:, None, None, None]
    ref = (ref ** 2).reshape(3, -1).sum(-1)
    wts = torch.tensor([1.0, -2.0, 0.5])
    (ref * wts).sum().backward()
    xd, bd = x.to(dev()).requires_grad_(True), bias.to(dev()).requires_grad_(True)
    h = G.bias_add_nchw(xd, bd)
    h = G.axpby(G.axpby(h, None, 2.0, 0.0, -1.0, 1.0 / np.sqrt(2.)), z.to(dev()))
    out = G.sumsq_rows(G.scale_rows(h, sc.to(dev()), divide=True))
    (out * wts.to(dev())).sum().backward()
    assert rel(out, ref) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(bd.grad, br.grad) < 1e-5


def test_dropout_mask_and_backward():
    from conditional_score_diffusion_amd import grad_ops as G
    x = torch.ones(4, 32, 16, 16, device=dev(), requires_grad=True)
    y = G.dropout(x, 0.1, 1234, 7)
    assert torch.equal(y, G.dropout(x, 0.1, 1234, 7))                 # counter-based: same key -> same mask
    assert not torch.equal(y, G.dropout(x, 0.1, 1234, 8))
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01
    assert torch.allclose(y[y != 0], torch.tensor(1 / 0.9, device=dev()))
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.allclose(x.grad, g * y.detach())


def _hip_loss_and_grads(case, precision='fp32', layout='nhwc', executor='planned', dropout=None):
    from conditional_score_diffusion_amd import losses
    cfg, B, x, y, u, tape = cases.grad_case(case)
    if dropout is not None:
        cfg.model.dropout = dropout
    cfg, nc, p, model = build(cfg, precision)
    model.train_layout = layout
    model.train_executor = executor
    sde = sdes_for(cfg)
    if cfg.model.name == 'ddpm':
        fn, batch = losses.get_general_sde_loss_fn(sde, True, False, True, True, True), x.to(dev())
    else:
        fn, batch = losses.get_general_sde_loss_fn(sde, True, True, True, True, True), (y.to(dev()), x.to(dev()))
    it = iter(tape)
    o_rand, o_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: u.clone()
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        loss = fn(model, batch)
    finally:
        torch.rand, torch.randn_like = o_rand, o_like
    assert model.training and loss.requires_grad
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in model.named_parameters()}, model


@pytest.mark.parametrize('layout,executor', [('nhwc', 'planned'), ('nhwc', 'operators'), ('nchw', 'operators')])
@pytest.mark.parametrize('case', list(cases.CASES))
def test_training_loss_and_grads_vs_reference(golden_dir, case, layout, executor):
    """loss.backward() through the HIP backward kernels vs the reference's autograd (fixture) and the oracle's (all entries);
    all three executors of the training graph: the planned graph behind csd_unet_train_forward / csd_unet_backward (default), autograd
    over the NHWC operators, autograd over the NCHW per-operator ABI."""
    g = np.load(os.path.join(golden_dir, 'grads.npz'))
    loss, grads, _ = _hip_loss_and_grads(case, layout=layout, executor=executor)
    worst = check_grads_vs_fixture(g, case, loss, grads, 1e-3)
    o_loss, o_grads = oracle_loss_and_grads(case)
    assert abs(loss - o_loss) <= 1e-4 * abs(o_loss)
    total = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in o_grads.values())))
    for k, v in o_grads.items():
        err = float((grads[k].cpu().double() - v.double()).abs().max())
        scale = max(float(v.abs().max()), float(v.double().norm()) / np.sqrt(v.numel()))
        assert err <= 1e-3 * scale + 1e-6 * total / np.sqrt(v.numel()), (k, err, scale)
    print(case, 'worst sampled rel err vs reference', worst)


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('fp16x3', 2e-4)])
@pytest.mark.parametrize('case', list(cases.CASES))
def test_planned_graph_equals_operator_graph(case, precision, tol):
    """csd_unet_train_forward / csd_unet_backward against autograd over the per-layer operators, dropout 0.1 ON (same Philox
    stream ids in both): same loss, same gradient for every parameter (the two differ only in summation order)."""
    la, ga, ma = _hip_loss_and_grads(case, precision, 'nhwc', 'planned', dropout=0.1)
    lb, gb, mb = _hip_loss_and_grads(case, precision, 'nhwc', 'operators', dropout=0.1)
    assert ma._train_calls == mb._train_calls == 1
    assert abs(la - lb) <= 1e-5 * abs(lb)
    total = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in gb.values())))
    worst = 0.0
    for k, v in gb.items():
        assert ga[k] is not None, k
        err = float((ga[k].double() - v.double()).abs().max())
        scale = max(float(v.abs().max()), float(v.double().norm()) / np.sqrt(v.numel()))
        assert err <= tol * scale + 1e-7 * total / np.sqrt(v.numel()), (k, err, scale)
        if scale > 1e-6 * total:           # (mathematically-zero gradients, e.g. the attention key bias, hold rounding noise only)
            worst = max(worst, err / scale)
    print(case, precision, 'planned vs operators: worst relative gradient difference %.2e' % worst)


def test_planned_graph_call_sequence_errors():
    """backward without a matching forward, a short workspace and NCSN++ (no planned training graph) fail loudly"""
    import ctypes
    from conditional_score_diffusion_amd._lib import lib, ptr
    cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
    cfg, nc, p, model = build(cfg)
    ps = model._train_params()
    table = (ctypes.c_void_p * len(ps))(*[q.data_ptr() for q in ps])
    gs = [torch.zeros_like(q) for q in ps]
    gtable = (ctypes.c_void_p * len(ps))(*[q.data_ptr() for q in gs])
    ws = model._train_workspace(B)
    dout = torch.zeros(B, model.out_channels, model.image_size, model.image_size, device=dev())
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B, 1, None) == -3     # CSD_ERR_STATE
    assert b'csd_unet_train_forward' in lib().csd_last_error()
    xs = x.to(dev()); ys = y.to(dev()); lab = torch.full((B,), 3., device=dev()); out = torch.empty_like(dout)
    assert lib().csd_unet_train_forward(model._h, table, ptr(ws), 1024, ptr(xs), ptr(ys), ptr(lab), ptr(out), B, 0.0, 0, 1, None) == -5
    assert lib().csd_unet_train_forward(model._h, table, ptr(ws), ws.numel(), ptr(xs), ptr(ys), ptr(lab), ptr(out), B, 0.0, 0, 1, None) == 0
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B + 1, 1, None) == -3
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B, 1, None) == 0
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B, 1, None) == -3        # consumed
    # a second forward into the SAME workspace before the first one's backward: that backward names call 2, the workspace holds call 3
    assert lib().csd_unet_train_forward(model._h, table, ptr(ws), ws.numel(), ptr(xs), ptr(ys), ptr(lab), ptr(out), B, 0.0, 0, 2, None) == 0
    assert lib().csd_unet_train_forward(model._h, table, ptr(ws), ws.numel(), ptr(xs), ptr(ys), ptr(lab), ptr(out), B, 0.0, 0, 3, None) == 0
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B, 2, None) == -3
    assert b're-used' in lib().csd_last_error()
    assert lib().csd_unet_backward(model._h, table, gtable, ptr(ws), ws.numel(), ptr(dout), B, 3, None) == 0
    torch.cuda.synchronize()


def test_two_live_forwards_of_one_model_keep_their_own_activations():
    """a monitoring forward (train mode, grad enabled) between a training forward and its backward must not disturb the first
    forward's saved activations: the second live forward gets its own workspace, both backwards give the gradients of their own
    inputs (one workspace per model used to be overwritten silently)"""
    cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
    cfg, nc, p, model = build(cfg)
    model.train()
    model._dropout = 0.0
    assert model.train_executor == 'planned'
    xs, ys = x.to(dev()), y.to(dev())
    lab = torch.full((B,), 3., device=dev())

    def grads_of(xin, other=None):
        for q in model.parameters():
            q.grad = None
        out = model({'x': xin, 'y': ys}, lab)
        if other is not None:
            out2 = model({'x': other, 'y': ys}, lab)          # alive at the same time as `out`
        out.square().sum().backward()
        g = torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()
        if other is not None:
            for q in model.parameters():
                q.grad = None
            out2.square().sum().backward()
            return g, torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()
        return g

    x2 = xs * 0.5 + 0.1
    ref1, ref2 = grads_of(xs), grads_of(x2)
    g1, g2 = grads_of(xs, other=x2)
    assert torch.equal(g1, ref1) and torch.equal(g2, ref2)
    assert not model._train_ws_busy
    assert torch.equal(grads_of(xs), ref1)                   # the shared workspace is free again


def test_stale_finalizer_does_not_release_the_next_forwards_workspace():
    """in an ordinary loop (`loss = f(model(x)); loss.backward()`) the graph of iteration N is freed only after iteration
    N + 1's forward has taken the shared workspace; the old context's finalizer must not hand that workspace to a monitoring forward
    that runs before backward N + 1 (ownership is keyed by the forward's call index)"""
    cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
    cfg, nc, p, model = build(cfg)
    model.train()
    model._dropout = 0.0
    xs, ys = x.to(dev()), y.to(dev())
    lab = torch.full((B,), 3., device=dev())

    def grads():
        return torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()

    for q in model.parameters():
        q.grad = None
    model({'x': xs, 'y': ys}, lab).square().sum().backward()
    ref = grads()
    loss = None
    for it in range(3):
        for q in model.parameters():
            q.grad = None
        loss = model({'x': xs, 'y': ys}, lab).square().sum()      # rebinding `loss` frees iteration it - 1's graph AFTER this forward
        assert model._train_ws_busy
        mon = model({'x': xs * 0.5, 'y': ys}, lab)                # monitoring forward before the backward: must get its own workspace
        assert model._train_ws_busy
        loss.backward()
        assert torch.equal(grads(), ref)
        del mon
    del loss
    import gc
    gc.collect()
    assert not model._train_ws_busy


def test_training_fp16x3_grads_close_to_fp32():
    """the split-fp16 convolutions in forward and data-gradient (weight gradient stays fp32 MFMA): fp32-class gradients"""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'grads.npz'))
    loss, grads, _ = _hip_loss_and_grads('sr3_tiny', 'fp16x3')
    check_grads_vs_fixture(g, 'sr3_tiny', loss, grads, 1e-3)


def test_sgd_steps_reduce_the_loss_with_dropout():
    """a few plain SGD steps on a fixed batch with dropout 0.1 active: loss is finite, gradients flow to every parameter,
    and the objective goes down."""
    from conditional_score_diffusion_amd import losses
    cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
    cfg.model.dropout = 0.1
    cfg, nc, p, model = build(cfg)
    sde = sdes_for(cfg)
    fn = losses.get_general_sde_loss_fn(sde, True, True, True, True, True)
    batch = (y.to(dev()), x.to(dev()))
    vals = []
    for step in range(4):
        torch.manual_seed(0)
        loss = fn(model, batch)
        model.zero_grad()
        loss.backward()
        assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())
        with torch.no_grad():
            for q in model.parameters():
                q -= 2e-4 * q.grad
        vals.append(float(loss.detach()))
    assert np.isfinite(vals).all() and vals[-1] < vals[0], vals
    model.eval()
    with torch.no_grad():
        out = model({'x': batch[1], 'y': batch[0]}, torch.full((B,), 500., device=dev()))     # planned executor, repacked weights
    assert torch.isfinite(out).all()


def test_fused_adam_ema_vs_torch():
    """csd_adam_step (clip + Adam + EMA in one pass over flat buffers) vs torch.optim.Adam + clip_grad_norm_ + the
    reference's EMA recurrence (losses.py:26-53, models/ema.py:61-90) on the host."""
    from conditional_score_diffusion_amd import optim
    rs = np.random.RandomState(3)
    shapes = [(7, 5), (33,), (4, 3, 3, 3), (129,)]
    ref = [torch.nn.Parameter(rnd(rs, *s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev())) for p in ref]
    topt = torch.optim.Adam(ref, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    flat = optim.FlatParams(mine)
    opt = optim.FusedAdam(flat, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ema = optim.ExponentialMovingAverage(flat, 0.999)
    shadow = [p.detach().clone() for p in ref]
    for step in range(1, 6):
        grads = [rnd(rs, *s) * (3.0 if step % 2 else 0.05) for s in shapes]      # norms above and below max_norm
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone()
        opt.zero_grad()
        for q, g in zip(mine, grads):
            q.grad.add_(g.to(dev()))
        torch.nn.utils.clip_grad_norm_(ref, max_norm=1.0)
        topt.step()
        decay = min(0.999, (1 + step) / (10 + step))
        for sp, p in zip(shadow, ref):
            sp.sub_((1 - decay) * (sp - p.detach()))
        opt.max_norm = 1.0
        opt.step(ema=ema)
        for p, q in zip(ref, mine):
            assert rel(q, p) < 2e-6
        got = [ema.shadow[int(o):int(o) + p.numel()].view_as(p) for p, o in zip(ref, flat.offsets[:-1])]
        for sp, gq in zip(shadow, got):
            assert rel(gq, sp) < 2e-6
    ema.store()
    ema.copy_to()
    assert rel(mine[0], shadow[0]) < 2e-6
    ema.restore()
    assert rel(mine[0], ref[0]) < 2e-6


def test_trainer_steps_single_process():
    """Trainer.train_step: loss -> HIP backward -> (1-rank) GradSync -> clip + Adam + EMA; the planned inference executor picks
    up the updated weights afterwards."""
    from conditional_score_diffusion_amd import train
    cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
    cfg.model.dropout = 0.1
    cfg.optim.warmup = 2
    cfg, nc, p, model = build(cfg)
    tr = train.Trainer(cfg, model, sdes_for(cfg))
    batch = (y.to(dev()), x.to(dev()))
    w0 = tr.flat.data.clone()
    vals = [float(tr.train_step(batch)) for _ in range(4)]
    assert np.isfinite(vals).all()
    assert tr.optimizer.num_steps == 4 and tr.step == 4
    assert float(tr.optimizer.last_grad_norm) > 0
    assert not torch.equal(tr.flat.data, w0)                       # (step 0 has warm-up factor 0; later steps move the weights)
    assert not torch.equal(tr.ema.shadow, tr.flat.data)
    ev = float(tr.eval_loss(batch))
    ev_ema = float(tr.eval_loss(batch, use_ema=True))
    assert np.isfinite([ev, ev_ema]).all()


def test_nhwc_operator_backward_vs_torch():
    """the NHWC forms (grad_ops_nhwc) of conv (3x3, stride 2, nearest-x2, NCHW-in / NCHW-out ends), GroupNorm+act, packed
    attention and the broadcast add against torch autograd"""
    from conditional_score_diffusion_amd import grad_ops_nhwc as G
    rs = np.random.RandomState(12)
    to_nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    to_nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
    for (B, Cin, Cout, H, k, stride, up2) in [(2, 32, 64, 12, 3, 1, False), (2, 64, 32, 8, 1, 1, False), (2, 32, 32, 16, 3, 2, False),
                                               (2, 32, 64, 8, 3, 1, True), (3, 96, 40, 5, 3, 1, False)]:
        x, w, b = rnd(rs, B, Cin, H, H), rnd(rs, Cout, Cin, k, k) * 0.1, rnd(rs, Cout)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        u = F.interpolate(xr, scale_factor=2, mode='nearest') if up2 else xr
        ref = F.conv2d(F.pad(u, (0, 1, 0, 1)), wr, br, stride=2) if stride == 2 else F.conv2d(u, wr, br, padding=k // 2)
        dy = rnd(rs, *ref.shape)
        ref.backward(dy)
        xd, wd, bd = to_nhwc(x).to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
        out = G.conv2d(xd, wd, bd, stride=stride, downsample_pad=stride == 2, up2=up2)
        assert rel(to_nchw(out), ref) < 1e-5
        out.backward(to_nhwc(dy).to(dev()))
        assert rel(to_nchw(xd.grad), xr.grad) < 1e-5 and rel(wd.grad, wr.grad) < 1e-5 and rel(bd.grad, br.grad) < 1e-5
    # the two ends of the network: NCHW in -> NHWC out (Cin = 6), NHWC in -> NCHW out (Cout = 3)
    x, w, b = rnd(rs, 2, 6, 12, 12), rnd(rs, 32, 6, 3, 3) * 0.1, rnd(rs, 32)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(x, wr, br, padding=1)
    dy = rnd(rs, *ref.shape)
    ref.backward(dy)
    wd, bd = w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    out = G.conv2d(x.to(dev()), wd, bd, layout=G.OUT_NHWC)
    out.backward(to_nhwc(dy).to(dev()))
    assert rel(to_nchw(out), ref) < 1e-5 and rel(wd.grad, wr.grad) < 1e-5 and rel(bd.grad, br.grad) < 1e-5
    x, w, b = rnd(rs, 2, 32, 12, 12), rnd(rs, 3, 32, 3, 3) * 0.1, rnd(rs, 3)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(xr, wr, br, padding=1)
    dy = rnd(rs, *ref.shape)
    ref.backward(dy)
    xd, wd, bd = to_nhwc(x).to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    out = G.conv2d(xd, wd, bd, layout=G.IN_NHWC)
    out.backward(dy.to(dev()))
    assert rel(out, ref) < 1e-5 and rel(to_nchw(xd.grad), xr.grad) < 1e-5 and rel(wd.grad, wr.grad) < 1e-5 and rel(bd.grad, br.grad) < 1e-5
    # GroupNorm + SiLU
    for (B, C, H, groups, act) in [(2, 32, 8, 32, 'swish'), (3, 96, 12, 32, 'swish'), (2, 64, 5, 32, 'none'), (2, 128, 16, 32, 'swish')]:
        x, ga, be = rnd(rs, B, C, H, H) * 2 + 0.3, rnd(rs, C) * 0.5 + 1, rnd(rs, C) * 0.2
        xr, gr, br = (t.clone().requires_grad_(True) for t in (x, ga, be))
        ref = F.group_norm(xr, groups, gr, br, eps=1e-6)
        ref = F.silu(ref) if act == 'swish' else ref
        dy = rnd(rs, *ref.shape)
        ref.backward(dy)
        xd, gd, bd = to_nhwc(x).to(dev()).requires_grad_(True), ga.to(dev()).requires_grad_(True), be.to(dev()).requires_grad_(True)
        out = G.groupnorm_act(xd, gd, bd, groups, 1e-6, act)
        assert rel(to_nchw(out), ref) < 1e-5
        out.backward(to_nhwc(dy).to(dev()))
        assert rel(to_nchw(xd.grad), xr.grad) < 2e-5 and rel(gd.grad, gr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5
    # packed attention
    for (B, C, H) in [(2, 32, 4), (2, 64, 8), (1, 96, 10)]:
        q, k, v = rnd(rs, B, C, H, H), rnd(rs, B, C, H, H), rnd(rs, B, C, H, H)
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
        wgt = torch.einsum('bchw,bcij->bhwij', qr, kr) * (int(C) ** (-0.5))
        wgt = F.softmax(wgt.reshape(B, H, H, H * H), dim=-1).reshape(B, H, H, H, H)
        ref = torch.einsum('bhwij,bcij->bchw', wgt, vr)
        do = rnd(rs, *ref.shape)
        ref.backward(do)
        qkv = torch.cat([to_nhwc(q), to_nhwc(k), to_nhwc(v)], dim=3).to(dev()).requires_grad_(True)
        out = G.attention(qkv)
        assert rel(to_nchw(out), ref) < 1e-5
        out.backward(to_nhwc(do).to(dev()))
        gq, gk, gv = (to_nchw(t) for t in qkv.grad.split(C, dim=3))
        assert rel(gq, qr.grad) < 2e-5 and rel(gk, kr.grad) < 2e-5 and rel(gv, vr.grad) < 2e-5
    # broadcast add
    x, bias = rnd(rs, 3, 16, 5, 5), rnd(rs, 3, 16)
    xd, bd = to_nhwc(x).to(dev()).requires_grad_(True), bias.to(dev()).requires_grad_(True)
    out = G.bias_add(xd, bd)
    assert rel(to_nchw(out), x + bias[:, :, None, None]) < 1e-6
    g = rnd(rs, 3, 5, 5, 16)
    out.backward(g.to(dev()))
    assert rel(bd.grad, g.sum(dim=(1, 2))) < 1e-5 and rel(xd.grad, g) == 0


@pytest.mark.parametrize('B,Cin,Cout,H,k', [(2, 32, 32, 16, 3), (3, 40, 72, 10, 3), (2, 64, 32, 8, 1), (4, 96, 96, 5, 3), (2, 128, 64, 20, 3),
                                            (1, 32, 32, 37, 3)])
def test_split_bf16_weight_gradient(B, Cin, Cout, H, k):
    """the weight gradient of the fp16 precision modes: split-bf16 operands on the bf16 matrix cores (wgrad_bf16.hip), both
    layouts, incl. maps narrower than a K block (two rows per wave-load) and widths that are not a multiple of 16"""
    from conditional_score_diffusion_amd import grad_ops as G, grad_ops_nhwc as GN
    rs = np.random.RandomState(21)
    x, w = rnd(rs, B, Cin, H, H), rnd(rs, Cout, Cin, k, k) * 0.1
    dy = rnd(rs, B, Cout, H, H) * 1e-4                      # small gradients: bf16 keeps the fp32 exponent range
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, padding=k // 2).backward(dy)
    wd = w.to(dev()).requires_grad_(True)
    G.conv2d(x.to(dev()), wd, None, precision='fp16x3').backward(dy.to(dev()))
    assert rel(wd.grad, wr.grad) < 5e-5
    wn = w.to(dev()).requires_grad_(True)
    GN.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev()), wn, None, precision='fp16x3').backward(dy.permute(0, 2, 3, 1).contiguous().to(dev()))
    assert rel(wn.grad, wr.grad) < 5e-5
    assert torch.equal(wn.grad, wd.grad)                     # same kernel, same reduction order


def test_weight_gradient_ab_schedules_agree(monkeypatch):
    """the A/B schedules kept behind environment switches (wide-load fp32 kernel, gather-only bf16 kernel) give the same weight
    gradient as the default ones"""
    from conditional_score_diffusion_amd import grad_ops_nhwc as GN
    rs = np.random.RandomState(31)
    x, w, dy = rnd(rs, 2, 20, 20, 64), rnd(rs, 96, 64, 3, 3) * 0.1, rnd(rs, 2, 20, 20, 96)

    def wgrad(precision):
        wd = w.to(dev()).requires_grad_(True)
        GN.conv2d(x.to(dev()), wd, None, precision=precision).backward(dy.to(dev()))
        return wd.grad.clone()

    base32, base16 = wgrad('fp32'), wgrad('fp16x3')
    monkeypatch.setenv('CSD_WGRAD_WIDE', '1')
    assert rel(wgrad('fp32'), base32) < 1e-5
    monkeypatch.delenv('CSD_WGRAD_WIDE')
    monkeypatch.setenv('CSD_WGRAD_GATHER', '1')
    assert rel(wgrad('fp16x3'), base16) < 1e-6          # same arithmetic, different data path
    monkeypatch.setenv('CSD_WGRAD_FP32', '1')
    assert torch.equal(wgrad('fp16x3'), base32)          # forced back to the exact fp32 kernel


@pytest.mark.parametrize('precision,tol', [('fp16x3', 1e-5), ('fp16', 3e-3)])
@pytest.mark.parametrize('B,Cin,Cout,H,stride,up2', [(2, 64, 96, 12, 1, False), (3, 128, 128, 16, 1, False), (2, 32, 128, 8, 2, False),
                                                     (2, 96, 256, 8, 1, True), (1, 256, 192, 5, 1, False)])
def test_nhwc_conv_on_the_quad_schedule(B, Cin, Cout, H, stride, up2, precision, tol):
    """NHWC convolutions with Cin % 32 == 0 and Cout % 96 == 0 or % 128 == 0 in the fp16 modes run the sampling path's quad-wave
    schedule (split pass + conv_f16_q_kernel, three or four 16-cout tiles per N half) - forward, and the data gradient through
    the transposed/flipped weight packing of that kernel"""
    from conditional_score_diffusion_amd import grad_ops_nhwc as G
    rs = np.random.RandomState(41)
    to_nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    to_nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
    x, w, b = rnd(rs, B, Cin, H, H), rnd(rs, Cout, Cin, 3, 3) * 0.05, rnd(rs, Cout)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    u = F.interpolate(xr, scale_factor=2, mode='nearest') if up2 else xr
    ref = F.conv2d(F.pad(u, (0, 1, 0, 1)), wr, br, stride=2) if stride == 2 else F.conv2d(u, wr, br, padding=1)
    dy = rnd(rs, *ref.shape)
    ref.backward(dy)
    xd, wd, bd = to_nhwc(x).to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    out = G.conv2d(xd, wd, bd, stride=stride, downsample_pad=stride == 2, up2=up2, precision=precision)
    assert rel(to_nchw(out), ref) < tol
    out.backward(to_nhwc(dy).to(dev()))
    assert rel(to_nchw(xd.grad), xr.grad) < tol
    assert rel(wd.grad, wr.grad) < max(tol, 5e-5)
    assert rel(bd.grad, br.grad) < 1e-5


@pytest.mark.parametrize('precision', ['fp16x3', 'fp32'])
def test_config4_shape_gradients_vs_oracle(precision):
    """BASELINE configs[3] shape (VS-CMDE edges2shoes: `ddpm_paired`, nf = 128, ch_mult (1,1,2,2), attention at 16 / 8, 64 x 64) at
    batch 2: the training loss and EVERY parameter gradient of the 28.75 M-parameter network against autograd over the oracle -
    the path the training bench times (NHWC graph, quad-schedule convolutions with four cout tiles, staged split-bf16 weight
    gradient on 64 / 32 / 16-wide maps and the two-rows-per-wave form on the 8-wide ones)."""
    from conditional_score_diffusion_amd import losses
    cfg = cases.make_config(name='ddpm_paired', nf=128, ch_mult=(1, 1, 2, 2), num_res_blocks=2, attn_resolutions=(16, 8), image_size=64,
                            sigma_max_y=float(np.sqrt(3 * 64 * 64)))
    cfg.model.dropout = 0.0
    cfg, nc, p, model = build(cfg, precision)
    assert sum(q.numel() for q in model.parameters()) == 28752902
    sde = sdes_for(cfg)
    B = 2
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(0, 1, size=(B, 3, 64, 64)).astype(np.float32))
    y = torch.from_numpy(rs.uniform(0, 1, size=(B, 3, 64, 64)).astype(np.float32))
    u = torch.tensor([0.71, 0.18])
    tape = cases.tape([(B, 3, 64, 64)] * 2, 9)
    fn = losses.get_general_sde_loss_fn(sde, True, True, True, True, True)
    it = iter(tape)
    o_rand, o_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: u.clone()
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    try:
        loss = fn(model, (y.to(dev()), x.to(dev())))
    finally:
        torch.rand, torch.randn_like = o_rand, o_like
    loss.backward()
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ve_x = so.VE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
    ve_y = so.VE(cfg.model.sigma_min_y, cfg.model.sigma_max_y, cfg.model.num_scales)
    t = u * (1 - 1e-5) + 1e-5
    ref = so.dsm_loss(pr, nc, 'ddpm_paired', ve_x, ve_y, x, y, t, tape[1], tape[0])
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-4 * abs(float(ref.detach()))
    total = float(np.sqrt(sum(float((v.grad.double() ** 2).sum()) for v in pr.values())))
    worst = 0.0
    rows = []
    for k, q in model.named_parameters():
        g, r = q.grad.cpu().double(), pr[k].grad.double()
        scale = max(float(r.abs().max()), float(r.norm()) / np.sqrt(r.numel()))
        err = float((g - r).abs().max())
        assert err <= 1e-3 * scale + 1e-6 * total / np.sqrt(r.numel()), (k, err, scale)
        worst = max(worst, err / max(scale, 1e-30))
        rows.append((err / max(scale, 1e-30), k, err, scale))
    # (tensors whose gradient is mathematically zero - the attention key bias NIN_1.b: softmax is invariant to a per-query shift -
    # hold rounding noise ~1e-11 of the norm; the bound above covers them through its absolute floor)
    big = [r_ for r_ in rows if r_[3] > 1e-6 * total]
    assert len(big) > 200 and max(r_[0] for r_ in big) < 1e-3
    print('config-4 shape, %s: worst relative gradient error over %d tensors %.2e' % (precision, len(big), max(r_[0] for r_ in big)))


def _ncsnpp_train_case():
    cfg, B, x, labels = cases.ncsnpp_case('ncsnpp_paired_skip')
    cfg.optim = cases.make_config().optim
    cfg.model.ema_rate = 0.999
    cfg.seed = 42
    cfg.training.likelihood_weighting, cfg.training.reduce_mean = True, True      # (the two-SDE loss branch requires it)
    cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.sigma_min_y, cfg.model.sigma_max_y = 0.01, 50., 0.01, 1.0
    return cfg, x


@pytest.mark.parametrize('family', ['ddpm', 'ncsnpp'])
def test_trainer_checkpoint_resume_is_bit_exact(tmp_path, family):
    """train 2 steps, save (torch.save of Trainer.state_dict()), train 2 more; a fresh model + Trainer restored from the file takes
    the same 2 steps: identical losses and identical parameters / EMA (dropout 0.1 on: the Philox stream position is part of the
    state - for the planned NCSN++ class too, whose training forward runs on a parameter-sharing twin)."""
    from conditional_score_diffusion_amd import sde_lib, train
    from conditional_score_diffusion_amd.models import utils as mutils
    if family == 'ncsnpp':
        cfg, xx = _ncsnpp_train_case()
        cfg.model.dropout = 0.1
        cfg.optim.warmup = 2
        batch = (xx[:, 3:].contiguous().to(dev()), xx[:, :3].contiguous().to(dev()))

        def build(cfg):
            torch.manual_seed(5)
            return None, None, None, mutils.create_model(cfg).to(dev())

        def sdes_for(cfg):
            return {'x': sde_lib.cVESDE(0.01, 50., 1000), 'y': sde_lib.VESDE(0.01, 1.0, 1000)}
    else:
        build, sdes_for = globals()['build'], globals()['sdes_for']
        cfg, B, x, y, u, tape = cases.grad_case('sr3_tiny')
        cfg.model.dropout = 0.1
        cfg.optim.warmup = 2
        batch = (y.to(dev()), x.to(dev()))

    def steps(tr, n):
        out = []
        for _ in range(n):
            torch.manual_seed(len(out) + tr.step)          # (t and the noise of the loss come from torch's generator)
            out.append(float(tr.train_step(batch)))
        return out

    _, _, _, model = build(cfg)
    tr = train.Trainer(cfg, model, sdes_for(cfg))
    steps(tr, 2)
    path = os.path.join(str(tmp_path), 'trainer.pt')
    torch.save(tr.state_dict(), path)
    cont = steps(tr, 2)
    _, _, _, model2 = build(cfg)
    tr2 = train.Trainer(cfg, model2, sdes_for(cfg))
    tr2.load_state_dict(torch.load(path))
    res = steps(tr2, 2)
    assert res == cont
    assert torch.equal(tr2.flat.data, tr.flat.data) and torch.equal(tr2.ema.shadow, tr.ema.shadow)
    assert tr2.step == tr.step == 4 and tr2.optimizer.num_steps == 4


def test_vs_cmde_trainer_shrinks_the_conditioning_sde():
    """VS-CMDE: with `reach_target_steps` / `sigma_max_y_target` in the config the Trainer rebuilds sde['y'] before every batch with the
    scheduled sigma_max_y (DecreasingVarianceConfigurationSetterCallback, lightning_callbacks/callbacks.py:23-78)."""
    from conditional_score_diffusion_amd import train
    cfg, B, x, y, u, tape = cases.grad_case('cmde_tiny')
    cfg.model.sigma_max_y = float(np.sqrt(3 * 20 * 20))
    cfg.model.reach_target_steps = 4
    cfg.model.sigma_max_y_target = 1.0
    cfg.model.sigma_min_y_target = cfg.model.sigma_min_y
    cfg, nc, p, model = build(cfg)
    tr = train.Trainer(cfg, model, sdes_for(cfg))
    batch = (y.to(dev()), x.to(dev()))
    seen = []
    for _ in range(5):
        loss = tr.train_step(batch)
        assert torch.isfinite(loss)
        seen.append(tr.sde['y'].sigma_max)
    f = train.get_reduction_fn(cfg.model.sigma_max_y, 4, 1.0)
    assert np.allclose(seen, [f(k) for k in range(5)]) and abs(seen[-1] - 1.0) < 1e-9
    assert tr.state_dict()['sigma_max_y'] == seen[-1]
