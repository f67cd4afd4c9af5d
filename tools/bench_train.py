"""Side bench (not the headline metric): training steps/s of BASELINE configs[3]-shaped work - VS-CMDE edges2shoes 64x64,
`ddpm_paired` nf=128, ch_mult (1,1,2,2), attention at 16/8, dropout 0.1, global batch 50 (configs/ve/inverse_problems/
image_to_image_translation/edges2shoes_ours_DV.py) - one full step = loss forward + HIP backward + bucketed gradient
all-reduce + fused clip/Adam/EMA.  One process per GPU (launch with torch.distributed.run for N > 1; the batch is sharded).

    python tools/bench_train.py [--steps 10 --warmup 3 --batch 50 --precision fp32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conditional_score_diffusion_amd import sde_lib, train  # noqa: E402
from conditional_score_diffusion_amd.config_dict import ConfigDict  # noqa: E402
from conditional_score_diffusion_amd.models import utils as mutils  # noqa: E402
import conditional_score_diffusion_amd.models.ddpm  # noqa: E402,F401

FWD_GFLOP_PER_IMAGE = 30.2          # SURVEY.md 8(d), cfg4 forward; a training step is ~3x (forward + dX + dW)


def make_config(precision):
    S = 64
    c = ConfigDict()
    c.training = ConfigDict(continuous=True, sde='vesde', likelihood_weighting=True, reduce_mean=True, batch_size=50)
    c.data = ConfigDict(image_size=S, effective_image_size=S, centered=False, shape_x=[3, S, S], shape_y=[3, S, S], num_channels=6)
    smax = float(np.sqrt(3 * S * S))
    c.model = ConfigDict(name='ddpm_paired', nf=128, ch_mult=(1, 1, 2, 2), num_res_blocks=2, attn_resolutions=(16, 8),
                         dropout=0.1, resamp_with_conv=True, conditional=True, nonlinearity='swish', num_scales=1000,
                         sigma_min_x=5e-3, sigma_max_x=smax, sigma_min_y=5e-3, sigma_max_y=smax, input_channels=6,
                         output_channels=6, embedding_type='positional', scale_by_sigma=True, ema_rate=0.999,
                         reach_target_steps=300000, sigma_max_y_target=1.0, sigma_min_y_target=5e-3,     # VS-CMDE schedule (edges2shoes_ours_DV.py:101-107)
                         csd_precision=precision)
    c.optim = ConfigDict(weight_decay=0, optimizer='Adam', lr=2e-4, beta1=0.9, eps=1e-8, warmup=2500, grad_clip=1)
    c.seed = 42
    return c


def make_ncsnpp_config(precision):
    """the same work shape on the architecture north_star names: ncsnpp_paired (BigGAN blocks, FIR resampling, input / output skips,
    skip_rescale) with the configs[3] hyper-parameters (64 x 64, nf = 128, ch_mult (1, 1, 2, 2), attention at 16, dropout 0.1)"""
    c = make_config(precision)
    m = c.model
    m.name = 'ncsnpp_paired'
    m.attn_resolutions = (16,)
    for k, v in dict(fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan', progressive='output_skip',
                     progressive_input='input_skip', progressive_combine='sum', attention_type='ddpm', init_scale=0.,
                     fourier_scale=16, conv_size=3, sigma_min=5e-3, sigma_max=m.sigma_max_x).items():
        setattr(m, k, v)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='ddpm_paired', choices=['ddpm_paired', 'ncsnpp_paired'])
    ap.add_argument('--executor', default=None, choices=['planned', 'operators'], help='training executor (default: the model\'s = planned)')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=50, help='global batch')
    ap.add_argument('--precision', default='fp32')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    grouped = 'RANK' in os.environ and 'MASTER_PORT' in os.environ      # launched by torch.distributed.run: RCCL also at world size 1
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if a.executor:
        os.environ['CSD_TRAIN_EXECUTOR'] = a.executor
    cfg = make_config(a.precision) if a.model == 'ddpm_paired' else make_ncsnpp_config(a.precision)
    import conditional_score_diffusion_amd.models.ncsnpp  # noqa: F401
    torch.manual_seed(0)
    model = mutils.create_model(cfg).to(dev)
    m = cfg.model
    sde = {'x': sde_lib.cVESDE(m.sigma_min_x, m.sigma_max_x, m.num_scales),
           'y': sde_lib.VESDE(m.sigma_min_y, m.sigma_max_y, m.num_scales)}
    tr = train.Trainer(cfg, model, sde)
    from conditional_score_diffusion_amd.distributed import shard_bounds
    lo, hi = shard_bounds(a.batch, rank, world)       # the config's global batch of 50 over 8 GPUs: 7,7,7,7,7,7,7,1 (ragged)
    B = hi - lo
    if B == 0:
        sys.exit('rank %d has no images (global batch %d over %d ranks)' % (rank, a.batch, world))
    global_n = a.batch if a.batch % world else None
    g = torch.Generator().manual_seed(1 + rank)
    batch = (torch.rand(B, 3, 64, 64, generator=g).to(dev), torch.rand(B, 3, 64, 64, generator=g).to(dev))

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = tr.train_step(batch, global_n=global_n)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_step(batch, global_n=global_n)
    sync()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if grouped:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank == 0:
        sps = a.steps / dt
        print(json.dumps({'metric': 'training steps/sec, VS-CMDE edges2shoes 64x64 (%s nf=128), fwd+bwd+all-reduce+Adam/EMA' % a.model,
                          'executor': getattr(model, 'train_executor', None),
                          'value': sps, 'unit': 'steps/sec', 'images_per_sec': sps * a.batch, 'n_gpus': world,
                          'global_batch': a.batch, 'rank0_batch': B, 'ms_per_step': 1e3 * dt / a.steps, 'steps': a.steps, 'warmup': a.warmup,
                          'precision': a.precision, 'params': tr.flat.numel, 'loss': float(loss),
                          'achieved_TFLOPs_3x_fwd': 3 * FWD_GFLOP_PER_IMAGE * 1e-3 * sps * a.batch, 'data': 'synthetic'}))
    if grouped:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
