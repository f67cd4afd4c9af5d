"""ORACLE tooling (build container only): run the IMPORTED REFERENCE on the seeded cases of
oracle/cases.py and write its outputs to tests/golden/*.npz.

    python oracle/make_goldens.py            # regenerates every fixture

The fixtures pin oracle/score_oracle.py (tests/test_oracle_golden.py) and are also compared
directly with the HIP path (tests/test_gpu_*.py).  Only inputs-by-seed and reference OUTPUTS are
stored - no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import ref_import  # noqa: E402
import score_oracle as so  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def build_ref_model(ref, cfg, seed=0):
    model = ref['models.utils'].create_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    # the oracle's own shape table must agree with the reference's state_dict exactly
    mine = so.ddpm_param_shapes(so.NetCfg.from_config(cfg))
    assert mine == shapes, set(mine.items()) ^ set(shapes.items())
    params = so.synth_params(shapes, seed)
    model.load_state_dict(params)
    model.eval()
    return model, params


def sdes_for(ref, cfg):
    sl = ref['sde_lib']
    m = cfg.model
    if m.name == 'ddpm_paired':
        return {'x': sl.cVESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales),
                'y': sl.VESDE(m.sigma_min_y, m.sigma_max_y, m.num_scales)}
    if m.name == 'ddpm':
        return sl.VESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales)
    return sl.cVESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales)


def gen_network_case(ref, case):
    cfg, B = cases.case_config(case)
    model, _ = build_ref_model(ref, cfg)
    sde = sdes_for(ref, cfg)
    y = cases.case_y(case)
    xs = (B,) + tuple(cfg.data.shape_x)
    out = {}
    rs = np.random.RandomState(7)
    name = cfg.model.name
    with torch.no_grad():
        # G3: whole-network forward + score at three times
        for j, tval in enumerate([1.0, 0.5, 1e-5]):
            sig = float(cfg.model.sigma_min_x * (cfg.model.sigma_max_x / cfg.model.sigma_min_x) ** tval)
            x = torch.from_numpy((rs.standard_normal(xs) * sig + 0.5).astype(np.float32))
            t = torch.ones(B) * tval
            labels = t * (cfg.model.num_scales - 1)
            if name == 'ddpm':
                # DDPM has no ``embedding_type`` attribute, which the unconditional continuous VESDE
                # branch reads (models/utils.py:251); set it on the instance (as NCSN++ defines it).
                model.embedding_type = 'positional'
                sfn = ref['models.utils'].get_score_fn(sde, model, conditional=False, train=False,
                                                       continuous=True)
                score = sfn(x, t)
                net = model(x, sde.marginal_prob(x, t)[1])   # label = sigma(t) in this branch
                out['net%d' % j] = net.numpy()
            else:
                net = model({'x': x, 'y': y}, labels)
                sfn = ref['models.utils'].get_score_fn(sde, model, conditional=True, train=False, continuous=True)
                sfn = ref['models.utils'].get_conditional_score_fn(sfn, target_domain='x')
                score = sfn(x, y, t)
                if isinstance(net, dict):
                    out['net%d' % j] = torch.cat([net['x'], net['y']], 1).numpy()
                else:
                    out['net%d' % j] = net.numpy()
            out['x%d' % j] = x.numpy()
            out['score%d' % j] = score.numpy()
        # G5/G6: PC trajectories with a noise tape
        for p_steps in (1, 10, 50):
            tp = cases.tape(cases.pc_tape_shapes(case, p_steps))
            with ref_import.TapeRandn(tp) as tr:
                if name == 'ddpm':
                    un = ref['sampling.unconditional']
                    pred = ref['sampling.predictors'].get_predictor('reverse_diffusion')
                    corr = ref['sampling.correctors'].get_corrector('langevin')
                    sampler = un.get_pc_sampler(sde, xs, pred, corr, snr=cfg.sampling.snr, p_steps=p_steps,
                                                c_steps=1, probability_flow=False, continuous=True,
                                                denoise=True, eps=1e-5)
                    res, info = sampler(model, show_evolution=(p_steps == 10))
                else:
                    co = ref['sampling.conditional']
                    pred = ref['sampling.predictors'].get_predictor(cfg.sampling.predictor)
                    corr = ref['sampling.correctors'].get_corrector(cfg.sampling.corrector)
                    sampler = co.get_pc_conditional_sampler(sde, xs, pred, corr, snr=cfg.sampling.snr,
                                                            p_steps=p_steps, c_steps=1, probability_flow=False,
                                                            continuous=True, denoise=True, use_path=False, eps=1e-5)
                    res, info = sampler(model, y, show_evolution=(p_steps == 10))
                assert tr.i == len(tp), (tr.i, len(tp))
            out['pc%d' % p_steps] = res.numpy()
            if p_steps == 10:
                ev = info['evolution']
                ev = ev['x'] if isinstance(ev, dict) else ev
                out['pc10_evolution'] = ev.numpy()
    np.savez_compressed(os.path.join(OUT, case + '.npz'), **out)
    print(case, {k: v.shape for k, v in out.items()})


def gen_modules(ref):
    """G1/G2: individual reference modules on seeded inputs (NCHW fp32)."""
    L = ref['models.layers']
    rs = np.random.RandomState(11)
    out = {}

    def rnd(*s):
        return torch.from_numpy(rs.standard_normal(s).astype(np.float32))

    act = torch.nn.SiLU()
    with torch.no_grad():
        for tag, cin, cout, hw in [('res_same', 32, 32, 10), ('res_proj', 96, 64, 5)]:
            blk = L.ResnetBlockDDPM(act, cin, cout, temb_dim=128, dropout=0.1).eval()
            shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
            blk.load_state_dict(so.synth_params(shapes, 3))
            x, temb = rnd(2, cin, hw, hw) * 2 + 0.3, rnd(2, 128)
            out[tag + '_x'], out[tag + '_temb'], out[tag + '_out'] = x.numpy(), temb.numpy(), blk(x, temb).numpy()
        for tag, c, hw in [('attn25', 64, 5), ('attn100', 32, 10)]:
            blk = L.AttnBlock(c).eval()
            shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
            blk.load_state_dict(so.synth_params(shapes, 4))
            x = rnd(2, c, hw, hw) * 1.5
            out[tag + '_x'], out[tag + '_out'] = x.numpy(), blk(x).numpy()
        for tag, mod, hw in [('down', L.Downsample(32, True), 10), ('up', L.Upsample(32, True), 5)]:
            mod = mod.eval()
            shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
            mod.load_state_dict(so.synth_params(shapes, 5))
            x = rnd(2, 32, hw, hw)
            out[tag + '_x'], out[tag + '_out'] = x.numpy(), mod(x).numpy()
        t = torch.tensor([999.0, 978.6124, 500.25, 0.00999], dtype=torch.float32)
        out['temb_t'], out['temb_out'] = t.numpy(), L.get_timestep_embedding(t, 96).numpy()
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **out)
    print('modules', {k: v.shape for k, v in out.items()})


def gen_sde_tables(ref):
    """G4: sigma tables, timestep-index sequences, per-t scalars of the VE SDEs."""
    sl = ref['sde_lib']
    out = {}
    smax = float(np.sqrt(3 * 160 * 160))
    sde = sl.cVESDE(5e-3, np.sqrt(np.prod([3, 160, 160])), 1000)
    out['discrete_sigmas'] = sde.discrete_sigmas.numpy()
    for p_steps in (50, 1000):
        ts = torch.linspace(sde.T, 1e-5, p_steps)
        out['timesteps%d' % p_steps] = ts.numpy()
        out['index%d' % p_steps] = (ts * (sde.N - 1) / sde.T).long().numpy()
        out['labels%d' % p_steps] = (ts * (sde.N - 1)).numpy()
        x = torch.zeros(p_steps, 1, 1, 1)
        out['G%d' % p_steps] = sde.discretize(x, ts)[1].numpy()
        out['std%d' % p_steps] = sde.marginal_prob(x, ts)[1].numpy()
        out['g%d' % p_steps] = sde.sde(x, ts)[1].numpy()
    vy = sl.VESDE(5e-3, 1.0, 1000)
    ts = torch.linspace(1, 1e-5, 8)
    x0, x1 = torch.ones(8, 1, 2, 2) * 0.3, torch.ones(8, 1, 2, 2) * 0.7
    m, s = vy.compute_backward_kernel(x0, x1, ts, torch.ones(8) * 0.02)
    out['bk_mean'], out['bk_std'] = m.numpy(), s.numpy()
    out['vy_g'] = vy.sde(x0, ts)[1].numpy()
    vp = sl.VPSDE(0.1, 20., 1000)
    out['vp_mean'], out['vp_std'] = [a.numpy() for a in vp.marginal_prob(x0, ts)]
    out['vp_f'], out['vp_G'] = [a.numpy() for a in vp.discretize(x0, ts)]
    out['vp_drift'], out['vp_diff'] = [a.numpy() for a in vp.sde(x0, ts)]
    assert abs(smax - sde.sigma_max) < 1e-9
    np.savez_compressed(os.path.join(OUT, 'sde_tables.npz'), **out)
    print('sde_tables', {k: v.shape for k, v in out.items()})


def step_score(x, t, y=None):
    """A closed-form stand-in for the score network (the update rules are what is being pinned): smooth in x,
    depends on t (and y), same formula in tests/test_gpu_steps.py."""
    tt = t.reshape(-1, 1, 1, 1)
    s = -(x - 0.25) / (1.0 + tt) + 0.1 * torch.sin(3.0 * x)
    if y is not None:
        s = s + 0.05 * y
    return s


# (ald on subVPSDE is not pinned: the reference class has no `alphas` and raises AttributeError)
# (probability-flow Euler-Maruyama is not pinned: the reference indexes a Python float there and raises TypeError)
STEP_CASES = [
    # name, sde class, sde kwargs, kind, registry name, probability_flow, conditional
    ('ve_em', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'euler_maruyama', False, False),
    ('ve_anc', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'ancestral_sampling', False, False),
    ('vp_em', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'euler_maruyama', False, False),
    ('vp_anc', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'ancestral_sampling', False, False),
    ('subvp_em', 'subVPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'predictor', 'euler_maruyama', False, False),
    ('cve_em', 'cVESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'predictor', 'conditional_euler_maruyama', False, True),
    ('ve_ald', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'corrector', 'ald', False, False),
    ('vp_ald', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'corrector', 'ald', False, False),
]
STEP_TIMES = [0.73, 0.31, 1e-4]     # (the last one is timestep 0: adjacent sigma = 0 in ancestral sampling)


def gen_steps(ref):
    """One update of every predictor / corrector the first round did not cover, run by the reference classes
    themselves with a noise tape -> tests/golden/steps.npz (inputs by seed, reference outputs)."""
    sl, pr, co = ref['sde_lib'], ref['sampling.predictors'], ref['sampling.correctors']
    g = torch.Generator().manual_seed(77)
    B, C, S = 3, 3, 8
    x0 = torch.randn(B, C, S, S, generator=g) * 2.0
    y0 = torch.rand(B, C, S, S, generator=g)
    z0 = torch.randn(2, B, C, S, S, generator=g)
    out = {'x0': x0.numpy(), 'y0': y0.numpy(), 'z0': z0.numpy(), 'times': np.array(STEP_TIMES, np.float32)}
    for name, scls, skw, kind, reg, pf, cond in STEP_CASES:
        sde = getattr(sl, scls)(**skw)
        for ti, tv in enumerate(STEP_TIMES):
            t = torch.full((B,), tv)
            if cond:
                score_fn = lambda x, y, t: step_score(x, t, y)
            else:
                score_fn = lambda x, t: step_score(x, t)
            with ref_import.TapeRandn([z0[0], z0[1]]):
                if kind == 'predictor':
                    obj = pr.get_predictor(reg)(sde, score_fn, pf)
                    x, xm = obj.update_fn(x0.clone(), y0, t) if cond else obj.update_fn(x0.clone(), t)
                else:
                    obj = co.get_corrector(reg)(sde, score_fn, 0.16, 2)
                    x, xm = obj.update_fn(x0.clone(), t)
            out['%s_t%d_x' % (name, ti)] = x.numpy()
            out['%s_t%d_xmean' % (name, ti)] = xm.numpy()
    np.savez_compressed(os.path.join(OUT, 'steps.npz'), **out)
    print('steps.npz: %d arrays' % len(out))


VP_LANGEVIN_CASES = [       # name, SDE class, kwargs, registry name, conditional
    ('vp_lang', 'VPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'langevin', False),
    ('ve_lang', 'VESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'langevin', False),
    ('cvp_lang', 'cVPSDE', dict(beta_min=0.1, beta_max=20., N=1000), 'conditional_langevin', True),
    ('cve_lang', 'cVESDE', dict(sigma_min=0.01, sigma_max=50., N=1000), 'conditional_langevin', True),
]


def gen_vp_langevin(ref):
    """Two Langevin corrector updates (sampling/correctors.py:51-108) of the reference classes on the VP SDEs (subVPSDE has no
    `alphas`: the reference raises AttributeError there, sde_lib.py:251-287; step size
    times alphas[timestep], :63-65,94-96) and, for contrast, the VE ones (alpha = 1), with a noise tape and the closed-form score
    of gen_steps -> tests/golden/vp_langevin.npz."""
    sl, co = ref['sde_lib'], ref['sampling.correctors']
    g = torch.Generator().manual_seed(79)
    B, C, S = 3, 3, 8
    x0 = torch.randn(B, C, S, S, generator=g) * 2.0
    y0 = torch.rand(B, C, S, S, generator=g)
    z0 = torch.randn(2, B, C, S, S, generator=g)
    out = {'x0': x0.numpy(), 'y0': y0.numpy(), 'z0': z0.numpy(), 'times': np.array(STEP_TIMES, np.float32)}
    for name, scls, skw, reg, cond in VP_LANGEVIN_CASES:
        sde = getattr(sl, scls)(**skw)
        for ti, tv in enumerate(STEP_TIMES):
            t = torch.full((B,), tv)
            score_fn = (lambda x, y, t: step_score(x, t, y)) if cond else (lambda x, t: step_score(x, t))
            with ref_import.TapeRandn([z0[0], z0[1]]):
                obj = co.get_corrector(reg)(sde, score_fn, 0.16, 2)
                x, xm = obj.update_fn(x0.clone(), y0, t) if cond else obj.update_fn(x0.clone(), t)
            out['%s_t%d_x' % (name, ti)] = x.numpy()
            out['%s_t%d_xmean' % (name, ti)] = xm.numpy()
            if scls != 'VESDE' and scls != 'cVESDE':
                out['%s_t%d_alpha' % (name, ti)] = np.float32(sde.alphas[(t[:1] * (sde.N - 1) / sde.T).long()].item())
    np.savez_compressed(os.path.join(OUT, 'vp_langevin.npz'), **out)
    print('vp_langevin.npz: %d arrays' % len(out))


def gen_use_path(ref):
    """use_path conditional PC sampling (sampling/conditional.py:124-178) of the reference on the CMDE tiny case, 4 steps,
    noise tape -> tests/golden/use_path.npz."""
    case = 'cmde_tiny'
    cfg, B = cases.case_config(case)
    model, _ = build_ref_model(ref, cfg)
    sde = sdes_for(ref, cfg)
    y = cases.case_y(case)
    xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
    P = 4
    shapes = [xs, ys] + [ys, xs, xs] * P              # prior, y_{T+tau}; per step: bridge y, predictor z, corrector z
    tp = cases.tape(shapes, seed=7)
    sc = ref['sampling.conditional']
    fn = sc.get_pc_conditional_sampler(sde, xs, ref['sampling.predictors'].get_predictor('conditional_reverse_diffusion'),
                                       ref['sampling.correctors'].get_corrector('conditional_langevin'), snr=cfg.sampling.snr,
                                       p_steps=P, c_steps=1, continuous=True, denoise=True, use_path=True, eps=1e-5)
    with ref_import.TapeRandn(tp) as tr:
        out, info = fn(model, y, show_evolution=True)
        assert tr.i == len(tp), (tr.i, len(tp))
    np.savez_compressed(os.path.join(OUT, 'use_path.npz'), out=out.numpy(), evo_x=info['evolution']['x'].numpy(),
                        evo_y=info['evolution']['y'].numpy())
    print('use_path', tuple(out.shape), float(out.abs().max()))


def gen_losses(ref):
    """Evaluation-loss values (losses.py:99-232) of the reference on the tiny cases with fixed t and noise -> tests/golden/losses.npz."""
    L = ref['losses']
    out = {}
    for case in cases.CASES:
        cfg, B = cases.case_config(case)
        model, _ = build_ref_model(ref, cfg)
        sde = sdes_for(ref, cfg)
        rs = np.random.RandomState(11)
        xs, ys = (B,) + tuple(cfg.data.shape_x), (B,) + tuple(cfg.data.shape_y)
        x = torch.from_numpy(rs.uniform(0, 1, size=xs).astype(np.float32))
        y = cases.case_y(case)
        tvals = torch.tensor([0.83, 0.21][:B])
        for lw in (True, False):
            if isinstance(sde, dict) and not lw:
                continue
            for rm in (True, False):
                if cfg.model.name == 'ddpm':
                    model.embedding_type = 'positional'      # (attribute the reference's score_fn reads; see gen_network_case)
                    fn, batch, tape = L.get_general_sde_loss_fn(sde, False, False, rm, True, lw), x, cases.tape([xs], 3)
                elif isinstance(sde, dict):
                    fn, batch, tape = L.get_general_sde_loss_fn(sde, False, True, rm, True, lw), (y, x), cases.tape([ys, xs], 3)
                else:
                    fn, batch, tape = L.get_general_sde_loss_fn(sde, False, True, rm, True, lw), (y, x), cases.tape([xs], 3)
                orig = torch.rand
                torch.rand = lambda *a, **k: tvals.clone()
                try:
                    with ref_import.TapeRandn(tape), torch.no_grad():
                        v = fn(model, batch)
                finally:
                    torch.rand = orig
                out['%s_lw%d_rm%d' % (case, lw, rm)] = np.float64(v.item())
                print(case, lw, rm, float(v))
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **out)


def gen_grads(ref):
    """Training loss (losses.py:99-232, train=True, dropout 0) and its parameter gradients from the reference's autograd on the
    tiny cases -> tests/golden/grads.npz: loss, and per parameter the gradient's L2 norm and its values at fixed entries."""
    L = ref['losses']
    out = {}
    for case in cases.CASES:
        cfg, B, x, y, tvals, tape = cases.grad_case(case)
        model, _ = build_ref_model(ref, cfg)
        sde = sdes_for(ref, cfg)
        if cfg.model.name == 'ddpm':
            model.embedding_type = 'positional'
            fn, batch = L.get_general_sde_loss_fn(sde, True, False, True, True, True), x
        else:
            fn, batch = L.get_general_sde_loss_fn(sde, True, True, True, True, True), (y, x)
        orig = torch.rand
        torch.rand = lambda *a, **k: tvals.clone()
        try:
            with ref_import.TapeRandn(tape):
                loss = fn(model, batch)
        finally:
            torch.rand = orig
        assert model.training
        loss.backward()
        out[case + '_loss'] = np.float64(loss.item())
        names, norms, samples = [], [], []
        for k, prm in model.named_parameters():
            g = prm.grad.detach().reshape(-1).double().numpy()
            names.append(k)
            norms.append(np.sqrt((g * g).sum()))
            idx = cases.grad_sample_index(k, g.size)
            sm = np.zeros(48)
            sm[:idx.size] = g[idx]
            samples.append(sm)
        out[case + '_names'] = np.array(names)
        out[case + '_norms'] = np.array(norms)
        out[case + '_samples'] = np.array(samples)
        print(case, 'loss', float(loss), 'grad norm', float(np.sqrt((np.array(norms) ** 2).sum())), len(names), 'tensors')
    np.savez_compressed(os.path.join(OUT, 'grads.npz'), **out)


def gen_ode(ref):
    """Probability-flow ODE sampler of the reference (sampling/unconditional.py:93-158: scipy RK45, rtol = atol = 1e-5, denoise) on
    the tiny unconditional case from a fixed latent -> tests/golden/ode.npz (final sample, number of function evaluations)."""
    cfg, B = cases.case_config('uncond_tiny')
    model, _ = build_ref_model(ref, cfg)
    model.embedding_type = 'positional'
    sde = sdes_for(ref, cfg)
    shape = (B,) + tuple(cfg.data.shape_x)
    z = cases.tape([shape], 17)[0] * float(cfg.model.sigma_max_x)
    sampler = ref['sampling.unconditional'].get_ode_sampler(sde, shape, denoise=True, eps=1e-5)
    x, nfe = sampler(model, z=z.clone())
    print('ode: nfe', nfe, 'max |x|', float(x.abs().max()))
    np.savez_compressed(os.path.join(OUT, 'ode.npz'), x=x.numpy(), nfe=np.int64(nfe))


def gen_inpaint(ref):
    """get_pc_inpainter of the reference (sampling/unconditional.py:230-345) on the tiny unconditional case -> tests/golden/inpaint.npz"""
    cfg, B, data, mask, tape = cases.inpaint_case()
    model, _ = build_ref_model(ref, cfg)
    model.embedding_type = 'positional'
    sde = ref['sde_lib'].VESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, 12)
    P, C = ref['sampling.predictors'], ref['sampling.correctors']
    fn = ref['sampling.unconditional'].get_pc_inpainter(sde, P.get_predictor('reverse_diffusion'), C.get_corrector('langevin'), snr=0.15,
                                                      n_steps=1, probability_flow=False, continuous=True, denoise=True, eps=1e-5)
    with ref_import.TapeRandn(tape):
        x, _ = fn(model, data, mask)
    print('inpaint: max |x|', float(x.abs().max()), 'known region err', float(((x - data) * mask).abs().max()))
    np.savez_compressed(os.path.join(OUT, 'inpaint.npz'), x=x.numpy())


def gen_legacy_losses(ref):
    """The discrete-time objectives (losses.py:236-265 SMLD, :320-340 DDPM) of the reference on the tiny unconditional net with fixed
    labels and noise -> tests/golden/legacy_losses.npz"""
    L, S = ref['losses'], ref['sde_lib']
    cfg, B = cases.case_config('uncond_tiny')
    model, _ = build_ref_model(ref, cfg)
    model.embedding_type = 'positional'
    rs = np.random.RandomState(11)
    xs = (B,) + tuple(cfg.data.shape_x)
    x = torch.from_numpy(rs.uniform(0, 1, size=xs).astype(np.float32))
    labels = torch.tensor([700, 123][:B])
    out = {}
    orig = torch.randint
    torch.randint = lambda *a, **k: labels.clone()
    try:
        for rm in (True, False):
            for lw in (True, False):
                fn = L.get_smld_loss_fn(S.VESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales), False, rm, lw)
                with ref_import.TapeRandn(cases.tape([xs], 3)), torch.no_grad():
                    out['smld_rm%d_lw%d' % (rm, lw)] = np.float64(fn(model, x).item())
            fn = L.get_ddpm_loss_fn(S.VPSDE(0.1, 20., cfg.model.num_scales), False, rm)
            with ref_import.TapeRandn(cases.tape([xs], 3)), torch.no_grad():
                out['ddpm_rm%d' % rm] = np.float64(fn(model, x).item())
    finally:
        torch.randint = orig
    print(out)
    np.savez_compressed(os.path.join(OUT, 'legacy_losses.npz'), **out)


def gen_ncsnpp(ref):
    """Reference NCSN++ forward (models/ncsnpp.py) on the seeded cases of cases.NCSNPP_CASES -> tests/golden/ncsnpp.npz:
    state_dict key order + shapes (as a string table) and the network output."""
    out = {}
    for case in cases.NCSNPP_CASES:
        cfg, B, x, labels = cases.ncsnpp_case(case)
        torch.manual_seed(0)
        model = ref['models.utils'].create_model(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(cases.ncsnpp_params(shapes, 5))
        model.eval()
        with torch.no_grad():
            if cfg.model.name == 'ncsnpp_paired':
                r = model({'x': x[:, :3], 'y': x[:, 3:]}, labels)
                y = torch.cat([r['x'], r['y']], dim=1)
            else:
                y = model(x, labels)
        out[case + '_out'] = y.numpy()
        out[case + '_keys'] = np.array(['%s|%s' % (k, ','.join(map(str, shapes[k]))) for k in model.state_dict()])
        print(case, tuple(y.shape), float(y.abs().max()), len(shapes), 'tensors')
    np.savez_compressed(os.path.join(OUT, 'ncsnpp.npz'), **out)


def gen_sr3_160_long(ref):
    """Full-size SR3-160 (BASELINE configs[1] network: nf = 96, ch_mult (1,1,2,2,3,3), attention at 20/10/5, 160 x 160), the REAL
    1000-step PC schedule (sampling/conditional.py:180-226), B = 2, noise from a seeded tape, run by the imported reference
    on the CPU -> tests/golden/sr3_160_long.npz: x after every 50th PC step on a stride-4 pixel grid, the full final x_mean,
    and one B = 1 network evaluation.  ~35 min on 8 cores."""
    import time
    cfg = cases.make_config(**cases.SR3_160)
    model, _ = build_ref_model(ref, cfg)
    sde = sdes_for(ref, cfg)
    P, B, every, st = cases.LONG_P, cases.LONG_B, cases.LONG_EVERY, cases.LONG_STRIDE
    y = cases.sr3_160_y(B)
    xs = (B, 3, 160, 160)
    out = {}
    with torch.no_grad():
        rs = np.random.RandomState(5)
        x1 = torch.from_numpy((rs.standard_normal((1, 3, 160, 160)) * 3.0 + 0.5).astype(np.float32))
        lab = torch.tensor([412.0])
        out['fwd_x'], out['fwd_label'] = x1.numpy(), lab.numpy()
        out['fwd_net'] = model({'x': x1, 'y': y[:1]}, lab).numpy()
    tp = cases.long_tape()
    co = ref['sampling.conditional']
    sampler = co.get_pc_conditional_sampler(sde, xs, ref['sampling.predictors'].get_predictor('conditional_reverse_diffusion'),
                                            ref['sampling.correctors'].get_corrector('conditional_langevin'),
                                            snr=cfg.sampling.snr, p_steps=P, c_steps=1, probability_flow=False, continuous=True,
                                            denoise=True, use_path=False, eps=1e-5)
    t0 = time.time()
    with ref_import.TapeRandn(tp) as tr:
        res, info = sampler(model, y, show_evolution=True)
        assert tr.i == len(tp), (tr.i, len(tp))
    ev = info['evolution']['x']                              # [P, B, 3, 160, 160]: x after every PC step
    out['steps'] = np.arange(every - 1, P, every)
    out['evo'] = ev[every - 1::every, :, :, ::st, ::st].contiguous().numpy()
    out['evo_absmax'] = ev[every - 1::every].abs().amax(dim=(1, 2, 3, 4)).numpy()
    out['evo_rms'] = ev[every - 1::every].double().pow(2).mean(dim=(1, 2, 3, 4)).sqrt().numpy()
    out['final'] = res.numpy()
    print('sr3_160_long: %.0f s; |x| max per snapshot' % (time.time() - t0), out['evo_absmax'], 'final absmax', float(res.abs().max()))
    np.savez_compressed(os.path.join(OUT, 'sr3_160_long.npz'), **out)


def _ref_pc_conditional(ref, cfg, model, sde, y, tape, p_steps, show_evolution=False):
    xs = (y.shape[0],) + tuple(cfg.data.shape_x)
    co = ref['sampling.conditional']
    sampler = co.get_pc_conditional_sampler(sde, xs, ref['sampling.predictors'].get_predictor(cfg.sampling.predictor),
                                            ref['sampling.correctors'].get_corrector(cfg.sampling.corrector), snr=cfg.sampling.snr,
                                            p_steps=p_steps, c_steps=1, probability_flow=False, continuous=True, denoise=True,
                                            use_path=False, eps=1e-5)
    with ref_import.TapeRandn(tape) as tr:
        res, info = sampler(model, y, show_evolution=show_evolution)
        assert tr.i == len(tape), (tr.i, len(tape))
    return res, info


def gen_sharded_modes(ref):
    """the two exactness modes of batch-sharded sampling on a global batch of 4 split over 2 ranks, 10 PC steps, one
    noise tape: 'global' = ONE reference process holding all 4 samples (batch-mean norms over 4), 'shard<r>' = the reference run
    independently on shard r (what Lightning-DDP testing does) with its slice of the tape -> tests/golden/sharded_modes.npz"""
    out = {}
    P, B = 10, 4
    for case in ('sr3_tiny', 'cmde_tiny'):
        cfg, _ = cases.case_config(case)
        model, _ = build_ref_model(ref, cfg)
        sde = sdes_for(ref, cfg)
        y = cases.case_y(case, B=B)
        tape = cases.tape(cases.pc_tape_shapes(case, P, B=B), seed=91)
        with torch.no_grad():
            res, _ = _ref_pc_conditional(ref, cfg, model, sde, y, tape, P)
            out[case + '_global'] = res.numpy()
            for r in range(2):
                rs, _ = _ref_pc_conditional(ref, cfg, model, sde, y[2 * r:2 * r + 2], [t[2 * r:2 * r + 2] for t in tape], P)
                out['%s_shard%d' % (case, r)] = rs.numpy()
        d = np.abs(np.concatenate([out[case + '_shard0'], out[case + '_shard1']]) - out[case + '_global']).max()
        print(case, 'global vs per-shard max difference', d)
    np.savez_compressed(os.path.join(OUT, 'sharded_modes.npz'), **out)


def gen_long_tiny(ref):
    """The REAL 1000-step schedule end to end on the tiny SR3 / CMDE nets (B = 2, noise tape by seed): x after every 100th PC
    step and the final denoised sample -> tests/golden/long_tiny.npz"""
    out = {}
    P = 1000
    for case in ('sr3_tiny', 'cmde_tiny'):
        cfg, B = cases.case_config(case)
        model, _ = build_ref_model(ref, cfg)
        sde = sdes_for(ref, cfg)
        y = cases.case_y(case)
        tape = cases.tape(cases.pc_tape_shapes(case, P), seed=1000)
        with torch.no_grad():
            res, info = _ref_pc_conditional(ref, cfg, model, sde, y, tape, P, show_evolution=True)
        out[case + '_final'] = res.numpy()
        out[case + '_evo'] = info['evolution']['x'][99::100].numpy()
        print(case, '1000 steps: |x| max', float(res.abs().max()))
    np.savez_compressed(os.path.join(OUT, 'long_tiny.npz'), **out)


def _import_eval_tools():
    """lightning_callbacks/evaluation_tools.py of the reference, imported unmodified; its missing THIRD-PARTY imports get stand-ins:
    matplotlib / PIL / torchvision are unused by the functions exercised here, and of OpenCV only ``getGaussianKernel`` and
    ``filter2D`` are called (by ``ssim``, which keeps the border-independent 'valid' region) - provided from their documented
    definitions on scipy (Gaussian of sigma s: exp(-(i-(n-1)/2)^2 / (2 s^2)) normalised; filter2D = correlation)."""
    import importlib.util
    import types
    import scipy.ndimage as ndi
    cv2 = types.ModuleType('cv2')

    def getGaussianKernel(n, sigma):
        i = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
        g = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (g / g.sum()).reshape(n, 1)

    cv2.getGaussianKernel = getGaussianKernel
    cv2.filter2D = lambda img, ddepth, kernel: ndi.correlate(img, kernel, mode='mirror')
    mods = {'cv2': cv2}
    for name in ('matplotlib', 'matplotlib.pyplot', 'PIL'):
        mods[name] = types.ModuleType(name)
    mods['matplotlib'].pyplot = mods['matplotlib.pyplot']
    ref_import.install()              # (torchvision / torchvision.transforms stand-ins)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location('_ref_eval_tools', os.path.join(ref_import.REF, 'lightning_callbacks', 'evaluation_tools.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def eval_case():
    """seeded image batches of the evaluation fixture: 'samples' = ground truth + noise, clamped (what the test callback compares)"""
    rs = np.random.RandomState(2025)
    x = rs.uniform(0, 1, size=(3, 3, 40, 48)).astype(np.float32)
    x = (x + np.roll(x, 1, axis=2) + np.roll(x, 1, axis=3) + np.roll(x, 2, axis=2)) / 4.0        # (some spatial correlation)
    s = np.clip(x + rs.standard_normal(x.shape).astype(np.float32) * 0.05, 0, 1)
    mask_info = np.array([[5, 7, 16], [0, 0, 20], [20, 28, 20]])
    return torch.from_numpy(x), torch.from_numpy(s), mask_info


def gen_eval(ref):
    """PSNR / SSIM / bicubic resize / consistency values of the reference's evaluation_tools on seeded images
    -> tests/golden/eval_tools.npz"""
    et = _import_eval_tools()
    x, s, mask_info = eval_case()
    nx = torch.swapaxes(x.clone(), 1, -1).numpy() * 255
    ns = torch.swapaxes(s.clone(), 1, -1).numpy() * 255
    out = {'mean_psnr': np.float64(et.calculate_mean_psnr(ns, nx)), 'mean_ssim': np.float64(et.calculate_mean_ssim(ns, nx)),
           'psnr_each': np.array([et.calculate_psnr(ns[i], nx[i]) for i in range(3)]),
           'ssim_each': np.array([et.calculate_ssim(ns[i], nx[i]) for i in range(3)])}
    for scale in (0.25, 0.125, 2.0):
        out['resize_%g' % scale] = et.resize(x, scale).numpy()
    out['consistency_sr'] = np.float64(et.get_calculate_consistency_fn('super-resolution')(s, x, 4))
    out['consistency_inp'] = np.float64(et.get_calculate_consistency_fn('inpainting')(s, x, mask_info))
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, 'eval_tools.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = ref_import.modules()
    if len(sys.argv) > 1:          # regenerate selected fixtures only: python oracle/make_goldens.py grads losses
        for name in sys.argv[1:]:
            globals()['gen_' + name](ref)
        return
    gen_grads(ref)
    gen_ode(ref)
    gen_inpaint(ref)
    gen_legacy_losses(ref)
    gen_sde_tables(ref)
    gen_modules(ref)
    gen_steps(ref)
    gen_vp_langevin(ref)
    gen_ncsnpp(ref)
    gen_use_path(ref)
    gen_losses(ref)
    for case in cases.CASES:
        gen_network_case(ref, case)


if __name__ == '__main__':
    main()
