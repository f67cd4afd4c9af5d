"""world_size-2 gloo test (CPU) of the N>1 sampling path: shard -> sample -> single all_gather.
The HIP sampler cannot run here (no GPU), so a deterministic stand-in sampler is injected; what is
under test is the sharding / seeding / gather logic of conditional_score_diffusion_amd.distributed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(model, y, seed=0, **kw):
    # deterministic function of (shard content, seed): lets the parent check placement and seeding
    return y * 2.0 + float(seed % 97), {'seed': seed, 'n': y.shape[0]}


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conditional_score_diffusion_amd import distributed as D
    y = torch.arange(8 * 3 * 2 * 2, dtype=torch.float32).reshape(8, 3, 2, 2)
    out, info = D.sample_sharded(_fake_sampler, None, y_global=y, seed=5)
    q.put((rank, out, info))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_and_seeds():
    from conditional_score_diffusion_amd import distributed as D
    assert [D.shard_bounds(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    with pytest.raises(ValueError):
        D.shard_bounds(10, 0, 4)
    assert len({D.rank_seed(42, r) for r in range(8)}) == 8


def test_two_rank_sharded_sampling_gloo():
    from conditional_score_diffusion_amd import distributed as D
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    y = torch.arange(8 * 3 * 2 * 2, dtype=torch.float32).reshape(8, 3, 2, 2)
    expect = torch.cat([y[:4] * 2 + float(D.rank_seed(5, 0) % 97), y[4:] * 2 + float(D.rank_seed(5, 1) % 97)])
    for rank, out, info in res:
        assert torch.equal(out, expect)            # every rank holds the whole batch, shards in rank order
        assert info['n'] == 4 and info['seed'] == D.rank_seed(5, rank)


def test_single_process_is_passthrough():
    from conditional_score_diffusion_amd import distributed as D
    y = torch.ones(4, 3, 2, 2)
    out, info = D.sample_sharded(_fake_sampler, None, y_global=y, seed=1)
    assert out.shape == y.shape and info['n'] == 4
