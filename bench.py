#!/usr/bin/env python
"""bench.py - images/sec for 1000-step PC sampling, SR3 CelebA-160 score network (BASELINE.json
configs[1]: configs/ve/inverse_problems/super_resolution/celebA_SR3_160.py values, batch 64 per
MI355X, random-init weights, synthetic low-resolution inputs).

    python bench.py --gpus N --steps K --warmup W

A "step" is one predictor-corrector iteration of the reference loop (Langevin corrector then
reverse-diffusion predictor = 2 score-network evaluations + 2 update kernels) over the whole
per-GPU batch, executed by the fused device loop (csd_pc_sample) with on-device Philox noise.
Every step of the 1000-step schedule costs the same (no data-dependent work), so
    value = images/sec of a full 1000-step sampling run = B_total / (1000 * seconds_per_step).
K consecutive steps of the real 1000-step schedule are timed (K = 1000 times the whole run).
Inputs are resident in HBM before the timed region.  One process per GPU; N > 1 is launched by
torch.distributed.run, the batch is sharded (64 images per GPU, weak scaling), no data-path
collective except the single all-gather of the finished samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: bf16/fp16 MFMA dense peak (not the 2:1-sparse figure)
# algorithmic work per image per network evaluation, SR3-160
ALG_FLOP_PER_IMG_NFE = 107.0e9
ALG_BYTES_PER_IMG_NFE = 923.5e6
ALG_WEIGHT_BYTES_PER_NFE = 173.9e6


def sr3_160_config():
    """The values of configs/ve/inverse_problems/super_resolution/celebA_SR3_160.py that the hot
    path reads (SURVEY.md section 5 'Config / flags')."""
    from conditional_score_diffusion_amd.config_dict import ConfigDict
    S = 160
    c = ConfigDict()
    c.training = ConfigDict(continuous=True, sde='vesde', likelihood_weighting=True, reduce_mean=True, conditioning_approach='sr3')
    c.sampling = ConfigDict(method='pc', predictor='conditional_reverse_diffusion', corrector='conditional_langevin',
                            n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.15)
    c.data = ConfigDict(image_size=S, effective_image_size=S, centered=False, shape_x=[3, S, S], shape_y=[3, S, S], num_channels=6)
    smax = float(np.sqrt(3 * S * S))
    c.model = ConfigDict(name='ddpm_paired_SR3', nf=96, ch_mult=(1, 1, 2, 2, 3, 3), num_res_blocks=2, attn_resolutions=(20, 10, 5),
                         dropout=0.1, resamp_with_conv=True, conditional=True, nonlinearity='swish', num_scales=1000,
                         sigma_min_x=5e-3, sigma_max_x=smax, sigma_min_y=5e-3, sigma_max_y=1.0, sigma_min=5e-3, sigma_max=smax,
                         input_channels=6, output_channels=3, embedding_type='positional', scale_by_sigma=True, ema_rate=0.999)
    c.optim = ConfigDict(weight_decay=0, optimizer='Adam', lr=2e-4, beta1=0.9, eps=1e-8, warmup=2500, grad_clip=1)
    c.seed = 42
    return c


_DTYPES = {'fp32': 'f32',
           'fp16x3': 'f16x3 (every operand carried as hi + lo fp16 = 22 significand bits, 3 MFMAs per product, f32 accumulate; f32-class: the per-op '
                     'test tolerance is the f32 kernel\'s, network error 1.9e-6 vs the reference)',
           'fp16f8': 'f16+f8 (operands split hi+lo; hi*hi on the fp16 MFMA, the two correction products with e4m3 operands on the fp8 MFMA, f32 '
                     'accumulate; ~15 significand bits per operand: narrower than the reference\'s f32)',
           'fp16': 'f16 (f32 accumulate; narrower than the reference\'s f32, not certified)'}


def _stable_hash(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def synth_weights(shapes, seed=0):
    """Random-init weights for the bench, generated HERE (the GPU leg imports nothing from oracle/): every >= 2-D tensor is
    U(+-sqrt(3 / fan_avg)) - the reference's ``default_init(1.0)`` (models/layers.py:54-91), also for its init_scale = 0 layers so
    that the sampler is not numerically degenerate (with the 1e-10 output layer the score is ~0 and the Langevin step
    size (snr |z| / |score|)^2 overflows; same FLOPs either way) - biases and GroupNorm affine parameters perturbed around 0 / 1."""
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        rs = np.random.RandomState((seed * 1000003 + _stable_hash(k)) % (2 ** 31 - 1))
        if len(shp) >= 2:
            if k.endswith('.W'):
                fan_in, fan_out = shp[0], shp[1]
            else:
                rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
                fan_in, fan_out = shp[1] * rf, shp[0] * rf
            lim = float(np.sqrt(3.0 / ((fan_in + fan_out) / 2.0)))
            v = rs.uniform(-lim, lim, size=shp)
        elif 'GroupNorm' in k and k.endswith('weight') or (k.count('.') == 2 and k.endswith('weight')):
            v = 1.0 + 0.1 * rs.standard_normal(shp)
        else:
            v = 0.05 * rs.standard_normal(shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def synth_y(B, seed=123):
    rs = np.random.RandomState(seed)
    lr = rs.uniform(0, 1, size=(B, 3, 20, 20)).astype(np.float32)
    return torch.from_numpy(np.repeat(np.repeat(lr, 8, axis=2), 8, axis=3))


def cpu_baseline(cfg, steps=4, B=4):
    """The CPU oracle (oracle/score_oracle.py, a validated port of the reference's PyTorch CPU path)
    on a bounded sample of the same workload: B images, `steps` PC iterations at 160x160.  The ONLY place this file touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import score_oracle as so
    nc = so.NetCfg.from_config(cfg)
    p = synth_weights(so.ddpm_param_shapes(nc), 0)
    y = synth_y(B)
    shapes = [(B, 3, 160, 160)] * (1 + 2 * steps)
    rs = np.random.RandomState(1)
    tape = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in shapes]
    cores = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.time()
        so.pc_sample_conditional(p, nc, y, so.NoiseTape(tape), (cfg.model.sigma_min_x, cfg.model.sigma_max_x), None,
                                 sr3=True, p_steps=1000, snr=cfg.sampling.snr, N=1000, max_steps=steps)
        dt = time.time() - t0
    return {'value': B / (1000.0 * dt / steps), 'unit': 'images/sec', 'cores': cores, 'host_logical_cores': os.cpu_count(),
            'host_cpu': _cpu_model(), 'kind': 'port',
            'sample': 'B=%d images x %d PC step(s) (=%d network evaluations) of the 1000-step schedule at 160x160, '
                      'torch %s CPU fp32, %d threads, %.1f s wall' % (B, steps, 2 * steps * B, torch.__version__, cores, dt)}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--precision', default=os.environ.get('CSD_PRECISION', 'fp16x3'), choices=['fp32', 'fp16x3', 'fp16f8', 'fp16'],
                    help='arithmetic of the 3x3 contractions; the default fp16x3 is fp32-class (operands carried as hi + lo fp16: 22 significand bits, '
                         'per-op tolerance = the fp32 kernel\'s); fp16f8 / fp16 are narrower than the reference\'s fp32 and reported as side figures only')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='tuning aid: time the loop without the in-library event profiler')
    ap.add_argument('--no-alt', action='store_true', help='skip the short side measurements of the other precision modes')
    ap.add_argument('--stub-sampler', action='store_true',
                    help='test aid (tests/test_distributed_cpu.py): run the launch / barrier / all_gather / MAX-over-ranks logic on the CPU with '
                         'the gloo backend and a stand-in for the HIP sampler (sleeps rank-dependent milliseconds per step); prints a reduced line')
    ap.add_argument('--cpu-thread-sweep', action='store_true', help='cpu_baseline leg only: seconds per B = 4 evaluation by thread count (no GPU work), then exit')
    ap.add_argument('--cpu-steps', type=int, default=50, help='PC steps of the CPU baseline sample (BASELINE.md section 4: B = 4, 50 steps)')
    args = ap.parse_args()

    if args.cpu_thread_sweep:      # evidence for the thread count of the cpu_baseline leg (profiles/r04_cpu_thread_sweep.txt)
        cfg = sr3_160_config()
        for th in (8, 16, 32, 64, 128, os.cpu_count() or 1):
            torch.set_num_threads(th)
            r = cpu_baseline(cfg, steps=1)
            print('threads %3d: %s' % (th, r['sample']), flush=True)
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)' % args.gpus)
    stub = args.stub_sampler
    if stub:
        dev = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            sys.exit('bench.py needs an MI355X: the HIP path has no CPU fallback')
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    # one process per GPU under torch.distributed.run: the RCCL process group exists whenever the launcher set RANK - also for a
    # world of ONE (`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`), so that the barrier / all_gather / MAX-reduce
    # branch below is the SAME code at N = 1 and at N = 8
    grouped = 'RANK' in os.environ and 'MASTER_PORT' in os.environ
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if stub:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    B = args.batch
    if stub:
        class _NoProfile:      # (the in-library event profiler belongs to the HIP library)
            @staticmethod
            def profile_select(*a): pass
            @staticmethod
            def profile_start(): pass
            @staticmethod
            def profile_stop(): return {'conv3x3': {'ms': 0.0, 'launches': 0, 'flops': 0.0, 'bytes': 0.0, 'alg_bytes': 0.0}}
        _lib = _NoProfile
        x = torch.full((B, 3, 8, 8), float(rank))

        def run_steps(first, n, seed):
            time.sleep(0.002 * (1 + rank) * n)       # rank r is (r + 1) x slower: the reported time must be the slowest rank's
            x.add_(float(n))
    else:
        from conditional_score_diffusion_amd import _lib, ops, sde_lib
        from conditional_score_diffusion_amd.models import utils as mutils
        from conditional_score_diffusion_amd.sampling import fused
        import ctypes

        cfg = sr3_160_config()
        cfg.model.csd_precision = args.precision
        B = args.batch
        torch.manual_seed(0)
        model = mutils.create_model(cfg)
        # random-init weights of that architecture; the reference's init_scale=0 layers are re-drawn at
        # scale 1 so the sampler is not numerically degenerate - same FLOPs either way
        model.load_state_dict(synth_weights({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0))
        model = model.to(dev).eval()
        sde = sde_lib.cVESDE(cfg.model.sigma_min_x, cfg.model.sigma_max_x, cfg.model.num_scales)
        y = synth_y(B, seed=123 + rank).to(dev)
        ts, labels, std_x, G, _ = fused.step_scalars(sde, 1000, 1e-5)

        model._ensure_packed()
        ws = model._workspace(B)
        scratch = torch.empty(_lib.lib().csd_pc_scratch_bytes(model._h, B), dtype=torch.uint8, device=dev)
        x = ops.randn((B, 3, 160, 160), 42 + rank, 0, dev)
        x = ops.scale_rows(x, torch.full((B,), float(sde.sigma_max), device=dev))

        def run_steps(first, n, seed):
            p = _lib.PCParams()
            p.n_steps = n
            f = lambda t: ctypes.cast(t[first:first + n].contiguous().data_ptr(), ctypes.POINTER(ctypes.c_float))  # noqa: E731
            keep = [labels[first:first + n].contiguous(), std_x[first:first + n].contiguous(), G[first:first + n].contiguous()]
            p.labels, p.std_x, p.G = [ctypes.cast(k.data_ptr(), ctypes.POINTER(ctypes.c_float)) for k in keep]
            p.std_y = None
            p.snr = float(cfg.sampling.snr)
            p.denoise = 0
            p.noise_tape = None
            p.seed = seed
            p.record = None
            _lib.check(_lib.lib().csd_pc_sample(model._h, _lib.ptr(model._packed), _lib.ptr(ws), ws.numel(),
                                                _lib.ptr(scratch), scratch.numel(), _lib.ptr(x), _lib.ptr(y), B,
                                                ctypes.byref(p), _lib.current_stream(dev)), 'pc_sample')

    def barrier():
        if grouped:
            torch.distributed.barrier()
        if not stub:
            torch.cuda.synchronize()

    K, W = args.steps, args.warmup
    if W > 0:
        run_steps(0, W, 1000 + rank)
    barrier()
    if not args.no_profile:
        # events around the dominant kernel class only, on every 8th step of the timed region: bracketing all ~1000
        # launches of every step costs ~5 ms of stream time per step (measured: the event records serialise launches),
        # which would be charged to `value`; the full per-class breakdown comes from a second, untimed region below
        PROF_STRIDE = 8
        _lib.profile_select(['conv3x3'], PROF_STRIDE)
        _lib.profile_start()
    t0 = time.perf_counter()
    failure = None
    try:
        run_steps(W, K, 2000 + rank)
    except FloatingPointError as e:      # NonFiniteError of csd_pc_sample's finiteness contract: carried across the group before the gather,
        failure = e                      # so that no rank is left waiting in the collective (round-5 advisor)
    if not stub:
        torch.cuda.synchronize()      # (this rank's steps are done: the per-rank figure below; csd_pc_sample has synchronised already)
    t_steps = time.perf_counter() - t0
    if grouped:
        bad = torch.tensor([1.0 if failure is not None else 0.0], device=dev)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if float(bad.item()) > 0:
            torch.distributed.destroy_process_group()
            raise failure if failure is not None else FloatingPointError('bench: the sampler state of another rank left the finite range')
    elif failure is not None:
        raise failure
    if grouped:     # the one collective of the sampling path: gather the finished samples
        out = torch.empty((world * B,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
        torch.distributed.all_gather_into_tensor(out, x)
        if not stub:
            torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0 - t_steps
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_stop()
    # untimed second region with every launch class bracketed: the per-class milliseconds (they include the event overhead)
    prof_all, prof_all_ms = prof, None
    if not args.no_profile:
        n2 = max(1, min(K, 5))
        _lib.profile_select(None, 1)
        barrier()
        _lib.profile_start()
        t1 = time.perf_counter()
        run_steps(W + K, n2, 3000 + rank)
        barrier()
        prof_all_ms = (time.perf_counter() - t1) / n2 * 1e3
        prof_all = _lib.profile_stop()
        prof_all = {k: dict(v, ms=v['ms'] * K / n2, launches=v['launches'] * K // n2, flops=v['flops'] * K / n2,
                            bytes=v['bytes'] * K / n2, alg_bytes=v['alg_bytes'] * K / n2) for k, v in prof_all.items()}
    # diagnostic for the first multi-GPU run (never part of `value`): every rank's own step time and what the all-gather cost it
    per_rank = {'ms_per_step': [t_steps / K * 1e3], 'all_gather_ms': [t_gather * 1e3]}
    if grouped:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([t_steps / K * 1e3, t_gather * 1e3], dtype=torch.float64, device=dev)
        every = torch.empty(world * 2, dtype=torch.float64, device=dev)
        torch.distributed.all_gather_into_tensor(every, mine)
        every = every.reshape(world, 2).cpu()
        per_rank = {'ms_per_step': [float(v) for v in every[:, 0]], 'all_gather_ms': [float(v) for v in every[:, 1]]}
    if stub:
        if rank == 0:
            gathered = out[:, 0, 0, 0].reshape(world, B)[:, 0].tolist() if grouped else [float(x[0, 0, 0, 0])]
            print(json.dumps({'metric': 'stub', 'value': B * world / (1000.0 * dt / args.steps), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
                              'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                              'config': {'workload': 'stub sampler (CPU, gloo)', 'images_per_gpu': B, 'global_batch': B * world},
                              'per_rank': per_rank, 'gathered_first_element_per_rank': gathered}))
        if grouped:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if not os.environ.get('CSD_LIB_PATH'):      # (tuning builds with ablated kernels produce garbage on purpose)
        assert torch.isfinite(x).all(), 'sampler state became non-finite'

    if rank == 0:
        ms_step = dt / K * 1e3
        total_images = B * world
        value = total_images / (1000.0 * dt / K)
        dom = prof['conv3x3']
        dom_tf = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
        kernel_ms = sum(v['ms'] for v in prof_all.values())
        # Roofline of the dominant kernel class, priced against the guide's peaks only (MI355X_MICROARCH.md: HBM3E 8 TB/s; dense MFMA
        # peak of the instruction the mode multiplies on): arithmetic intensity = ALGORITHMIC flops / ALGORITHMIC bytes of the class's
        # launches (SURVEY.md 8d definitions, counted by the library per launch); ridge = MFMA peak / HBM peak.  AI < ridge -> the class
        # is HBM-bound by the roofline model and frac = algorithmic bytes / event-measured duration / 8 TB/s; otherwise frac = algorithmic
        # flops / duration / MFMA peak.  What the split arithmetic costs on the matrix pipe (3 MFMAs per algorithmic product in fp16x3)
        # is charged to the kernel, not to the roof: it is reported separately under `matrix_pipe`.
        modes = {
            'fp32': dict(peak=F32_MFMA_PEAK_TF, mfma_per_product=1.0, operand='fp32 (24 significand bits)',
                         kernel='conv_f32_kernel (3x3 stride-1 implicit GEMM, v_mfma_f32_32x32x2_f32)'),
            # class 'conv3x3' in this mode = the conv_xk launches only (the 160^2 / 80^2 / 40^2 levels: 92 % of the 3x3 stride-1 flops of the
            # SR3-160 shape): 1-D Winograd F(2,3), 4 instead of 6 contractions per output pair, i.e. 3 x 2/3 = 2 MFMAs per algorithmic
            # product.  The first layer and the quad kernel below 40^2 report as 'conv3x3_other' (kernel_classes_ms_per_step)
            'fp16x3': dict(peak=F16_MFMA_PEAK_TF, mfma_per_product=2.0, operand='hi + lo fp16 per operand (22 significand bits; the lo*lo term, 2^-22 relative, is dropped)',
                           kernel='conv_xk_kernel: every 3x3 stride-1 convolution of the 160^2 / 80^2 / 40^2 levels (1-D Winograd F(2,3) along the row, one transform '
                                  'component per wave: fused GroupNorm+SiLU+transform+split prologue as fillers between the MFMAs of one software-pipelined stream '
                                  'per SIMD, weights from L2 straight into registers, persistent 4-wave workgroup per CU, 2x v_mfma_f32_32x32x16_f16 per '
                                  'algorithmic product; ragged 16x16 tiles at 40^2)'),
            'fp16f8': dict(peak=F16_MFMA_PEAK_TF, mfma_per_product=2.0, operand='fp16 hi x fp16 hi + two correction products with e4m3 operands (~15 significand bits per operand: '
                                                                                  'NARROWER than the reference\'s fp32)',
                           kernel='3x3 stride-1 convolution class: conv_ff_kernel<NS=2,F8> / conv_fx_kernel (hi*hi on v_mfma_f32_32x32x16_f16, corrections on '
                                  'v_mfma_scale_f32_32x32x64_f8f6f4) + conv_f16_q_kernel<NS=2>'),
            'fp16': dict(peak=F16_MFMA_PEAK_TF, mfma_per_product=1.0, operand='fp16 (11 significand bits: NARROWER than the reference\'s fp32, not certified)',
                         kernel='3x3 stride-1 convolution class: conv_ff_kernel<NS=1> + conv_f16_lc_kernel<NS=1>'),
        }[args.precision]
        dom_kernel, dom_peak = modes['kernel'], modes['peak']
        # the kernel's ALGORITHMIC bytes are input + output tensor of the layer in fp32 - nothing else.  What the kernel
        # itself has to move on top of that (the residual read of a block's second convolution) is reported beside it as
        # kernel_bytes_per_launch and is NOT part of `achieved` / `frac`
        dom_alg = dom.get('alg_bytes', dom['bytes'])
        dom_gbs = dom_alg / (dom['ms'] * 1e-3) / 1e9 if dom['ms'] > 0 else 0.0
        dom_ai = dom['flops'] / max(dom_alg, 1.0)
        ridge = dom_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        if dom_ai < ridge:
            roof = {'bound': 'hbm', 'achieved': dom_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dom_gbs / HBM_PEAK_GBS}
        else:
            roof = {'bound': 'mfma', 'achieved': dom_tf, 'peak': dom_peak, 'unit': 'TFLOP/s', 'frac': dom_tf / dom_peak}
        # HBM traffic of the same kernel class from PMC counters (a separate rocprofv3 pass cannot run inside this
        # process): the committed summary of tools/pmc_hbm.sh for this mode, bytes per launch like `achieved`
        traffic, traffic_src = None, None
        for rnd in ('r06', 'r05', 'r04', 'r03'):
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', '%s_hbm_traffic_%s.json' % (rnd, args.precision))
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj['conv3x3_class']['hbm_bytes_per_launch']
                traffic_src = 'profiles/' + os.path.basename(tpath) + ' (' + tj['source'] + ')'
                break
        roof.update({'kernel': dom_kernel, 'traffic': traffic, 'traffic_unit': 'bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)',
                     'traffic_source': traffic_src,
                     'algorithmic_bytes_per_launch': dom_alg / max(dom['launches'], 1),
                     'algorithmic_bytes_definition': 'SURVEY.md 8(d): (input + output tensor of the layer) x 4 B, summed over the launches of the class and '
                                                     'divided by their count; `achieved` = these bytes / the event-measured duration of the same launches',
                     'kernel_bytes_per_launch': dom['bytes'] / max(dom['launches'], 1),
                     'kernel_bytes_definition': 'what the kernel has to move: algorithmic bytes + the residual tensor the second convolution of a block reads',
                     'peak_note': 'guide peaks: HBM3E 8000 GB/s; dense MFMA %.1f TFLOP/s for this mode\'s matrix instruction' % dom_peak,
                     'arithmetic_intensity_flop_per_byte': dom_ai, 'ridge_flop_per_byte': ridge,
                     'achieved_TFLOPs': dom_tf, 'achieved_GBs': dom_gbs,
                     'matrix_pipe': {'mfma_per_algorithmic_product': modes['mfma_per_product'],
                                     'issued_TFLOPs': dom_tf * modes['mfma_per_product'], 'peak_TFLOPs': dom_peak,
                                     'frac': dom_tf * modes['mfma_per_product'] / dom_peak},
                     'launches': dom['launches'], 'avg_launch_ms': dom['ms'] / max(dom['launches'], 1),
                     'sampling': 'events on every 8th PC step of the timed region (dominant class only)' if not args.no_profile else 'off',
                     'share_of_kernel_time': prof_all['conv3x3']['ms'] / max(kernel_ms, 1e-9)})
        # north_star yardstick: HBM roofline of the whole sampling run
        bytes_per_img = 2000 * (ALG_BYTES_PER_IMG_NFE + ALG_WEIGHT_BYTES_PER_NFE / B)
        hbm_roof = HBM_PEAK_GBS * 1e9 / bytes_per_img                      # images/s/GPU
        flop_roof = F32_MFMA_PEAK_TF * 1e12 / (2000 * ALG_FLOP_PER_IMG_NFE)
        res = {
            'metric': 'images/sec for 1000-step PC sampling, ddpm_paired_SR3 score net of celebA_SR3_160 (BASELINE configs[1]), 160x160',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': _DTYPES[args.precision], 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: celebA_SR3_160 (ddpm_paired_SR3, nf=96, ch_mult (1,1,2,2,3,3), '
                                   'attn 20/10/5), 1000-step PC (reverse_diffusion + langevin, snr 0.15), '
                                   'batch %d per GPU, random-init weights, synthetic LR inputs' % B,
                       'images_per_gpu': B, 'global_batch': total_images, 'pc_steps_timed': K,
                       'nfe_per_step': 2, 'noise': 'on-device Philox4x32-10', 'precision_mode': args.precision},
            'per_rank': per_rank,
            'roofline': roof,
            'hbm_roofline': {'images_per_sec_per_gpu': hbm_roof, 'frac': value / world / hbm_roof,
                             'achieved_GBs': value / world * bytes_per_img / 1e9, 'peak_GBs': HBM_PEAK_GBS},
            'flop_roofline_f32': {'images_per_sec_per_gpu': flop_roof, 'frac': value / world / flop_roof,
                                  'achieved_TFLOPs': value / world * 2000 * ALG_FLOP_PER_IMG_NFE / 1e12},
            'kernel_classes_ms_per_step': {k: v['ms'] / K for k, v in prof_all.items()},
            'kernel_classes_note': 'second, untimed region of %d steps with every launch bracketed by events (%s ms per step '
                                   'there; the event records themselves cost ~5 ms per step)' % (max(1, min(K, 5)), ('%.2f' % prof_all_ms) if prof_all_ms else 'n/a'),
            'kernel_time_fraction_of_wall': kernel_ms / ((prof_all_ms or (dt / K * 1e3)) * K),
        }
        if not args.no_cpu_baseline and world == 1:
            # oneDNN conv scaling collapses past ~16-32 threads on this host (profiles/r04_cpu_thread_sweep.txt, tools/cpu_threads.py:
            # seconds per B = 4 evaluation by thread count), so the baseline uses the best setting, not all cores
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            res['cpu_baseline'] = cpu_baseline(cfg, steps=args.cpu_steps)
        else:
            res['cpu_baseline'] = None
        if world == 1 and not args.no_alt:
            # the other two arithmetic modes, measured the same way in short side runs (reported, not the headline)
            import subprocess
            alt = {}
            for mode in ('fp32', 'fp16x3', 'fp16f8', 'fp16'):
                if mode == args.precision:
                    continue
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--precision', mode, '--steps', '10',
                                        '--warmup', '2', '--no-cpu-baseline', '--no-alt', '--batch', str(B)],
                                       capture_output=True, text=True, timeout=600)
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    alt[mode] = {'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'dtype': j['dtype'],
                                 'hbm_roofline_frac': j['hbm_roofline']['frac'], 'steps': 10,
                                 'creditable_as_headline': mode in ('fp32', 'fp16x3')}
                except Exception as e:      # a side measurement must never break the bench line
                    alt[mode] = {'error': str(e)[:200]}
            res['other_precision_modes'] = alt
            # what ONE GPU can show of strong scaling: the same loop at the per-GPU batch of a global batch of 64 sharded
            # over 8 / 4 / 2 GPUs.  predictor(N) = N x value(B = 64 / N) / value(B = 64): the speed-up N GPUs would reach on a global batch
            # of 64 with a free all-gather.  (bench.py --gpus N itself is WEAK scaling: 64 images per GPU.)
            try:
                import subprocess
                if B != 64:
                    raise ValueError('measured for the default per-GPU batch (64) only')
                ss = {}
                for bb in (8, 16, 32):
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--precision', args.precision, '--steps', '20', '--warmup', '5',
                                        '--no-cpu-baseline', '--no-alt', '--no-profile', '--batch', str(bb)], capture_output=True, text=True, timeout=600)
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    ss['B%d' % bb] = {'value': j['value'], 'ms_per_step': j['ms_per_step']}
                res['strong_scaling_predictor'] = dict(
                    ss, global_batch=64, value_B64=value,
                    predicted_speedup={'8_gpus': 8 * ss['B8']['value'] / value, '4_gpus': 4 * ss['B16']['value'] / value,
                                       '2_gpus': 2 * ss['B32']['value'] / value},
                    note='one-GPU measurement of the per-GPU share of a global batch of 64; no communication term (the single all-gather of '
                         '64 x 3 x 160 x 160 floats is < 0.1 ms over xGMI)')
            except Exception as e:
                res['strong_scaling_predictor'] = {'error': str(e)[:200]}
            # SURVEY 8(d) "config 4": one data-parallel training step (loss + HIP backward + gradient all-reduce + fused
            # clip/Adam/EMA) of the VS-CMDE edges2shoes-64 shape, measured by tools/bench_train.py - a side figure, not the headline
            try:
                tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'bench_train.py')
                r = subprocess.run([sys.executable, tool, '--precision', 'fp16x3', '--steps', '10', '--warmup', '3'],
                                   capture_output=True, text=True, timeout=600)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                res['training_side_bench'] = {k: j[k] for k in ('metric', 'value', 'unit', 'images_per_sec', 'global_batch', 'ms_per_step',
                                                                'precision', 'params', 'achieved_TFLOPs_3x_fwd')}
                # what ONE GPU of an 8-way data-parallel run of this config executes: 50 images over 8 ranks = 7 per GPU (ragged tail 1)
                r = subprocess.run([sys.executable, tool, '--precision', 'fp16x3', '--steps', '10', '--warmup', '3', '--batch', '7'],
                                   capture_output=True, text=True, timeout=600)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                res['training_side_bench']['per_gpu_batch_7_of_8way_dp'] = {'steps_per_sec': j['value'], 'ms_per_step': j['ms_per_step'],
                                                                            'images_per_sec': j['images_per_sec']}
                # the same step on the architecture north_star names (ncsnpp_paired, planned training graph, csrc/train_graph.h arch 1)
                r = subprocess.run([sys.executable, tool, '--precision', 'fp16x3', '--steps', '10', '--warmup', '3', '--model', 'ncsnpp_paired'],
                                   capture_output=True, text=True, timeout=600)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                res['training_side_bench']['ncsnpp_paired_same_shape'] = {'steps_per_sec': j['value'], 'ms_per_step': j['ms_per_step'],
                                                                          'images_per_sec': j['images_per_sec'], 'params': j['params'],
                                                                          'executor': j['executor']}
            except Exception as e:
                res['training_side_bench'] = {'error': str(e)[:200]}
            # BASELINE configs[2] (CMDE inpainting 128x128, two SDEs) and the architecture north_star names (NCSN++ with the SR3-160
            # hyper-parameters), same precision mode, B = 64: side figures with their own HBM-roofline fractions (SURVEY.md 8d bytes)
            try:
                tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'bench_other.py')
                r = subprocess.run([sys.executable, tool, args.precision, 'bench'], capture_output=True, text=True, timeout=600)
                js = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith('{')]
                cm, nc, n256 = js[0], js[1], js[2]
                n256f = js[3] if len(js) > 3 else None      # the same network with the chip filled (B = 32): per-GPU batch 8 leaves the 256 CUs short of tiles
                n256_bytes = 3645.7e6 + 4.0 * n256['params'] / n256['batch']      # fp32 layer-granular bytes per image-evaluation
                cm_roof = HBM_PEAK_GBS * 1e9 / (2000 * (591.3e6 + ALG_WEIGHT_BYTES_PER_NFE / 64))
                nc_bytes = 1085.7e6 + 4.0 * nc['params'] / 64
                res['side_benches'] = {
                    'configs2_cmde_inpainting_128': {'images_per_sec_1000_steps': cm['images_per_sec_1000_steps'], 'ms_per_pc_step': cm['ms_per_pc_step'],
                                                     'batch': cm['batch'], 'hbm_roofline_images_per_sec': cm_roof,
                                                     'hbm_roofline_frac': cm['images_per_sec_1000_steps'] / cm_roof},
                    'ncsnpp_paired_sr3_160_hyperparameters': {'image_evaluations_per_sec': nc['images_per_sec_per_nfe'], 'batch': nc['batch'],
                                                              'images_per_sec_1000_steps_equiv': nc['images_per_sec_per_nfe'] / 2000.0,
                                                              'params': nc['params'], 'algorithmic_bytes_per_image_nfe': nc_bytes,
                                                              'hbm_roofline_frac': nc['images_per_sec_per_nfe'] * nc_bytes / (HBM_PEAK_GBS * 1e9)},
                    # BASELINE configs[4] (NCSN++ 256^2).  The config names "fp16 with fp32 GroupNorm accumulate": fp16 ACTIVATIONS cannot
                    # hold the 1e-3 tolerance on this net (measured 1.2e-3 norm-wise / 4.5e-3 element-wise) - the certified replacement is
                    # this run's mode (fp32 residual stream, split fp16(+fp8) operands rebuilt per conv, fp32 / fp64 GroupNorm statistics)
                    'configs4_ncsnpp_256': {'image_evaluations_per_sec': n256['images_per_sec_per_nfe'], 'batch': n256['batch'],
                                            'images_per_sec_2000_step_pc_equiv': n256['images_per_sec_per_nfe'] / 4000.0,
                                            'params': n256['params'], 'algorithmic_bytes_per_image_nfe': n256_bytes,
                                            'hbm_roofline_frac': n256['images_per_sec_per_nfe'] * n256_bytes / (HBM_PEAK_GBS * 1e9)}}
                if n256f:
                    fb = 3645.7e6 + 4.0 * n256f['params'] / n256f['batch']
                    res['side_benches']['configs4_ncsnpp_256']['filled'] = {
                        'image_evaluations_per_sec': n256f['images_per_sec_per_nfe'], 'batch': n256f['batch'],
                        'hbm_roofline_frac': n256f['images_per_sec_per_nfe'] * fb / (HBM_PEAK_GBS * 1e9)}
            except Exception as e:
                res['side_benches'] = {'error': str(e)[:200]}
        print(json.dumps(res))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
