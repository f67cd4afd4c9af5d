"""world_size-2 gloo test (CPU) of the N>1 sampling path: shard -> sample -> single all_gather.
The HIP sampler cannot run here (no GPU), so a deterministic stand-in sampler is injected; what is
under test is the sharding / seeding / gather logic of conditional_score_diffusion_amd.distributed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


_RENDEZVOUS_MARKERS = ('address already in use', 'eaddrinuse', 'connection refused', 'connection reset', 'failed to connect',
                       'socket timeout', 'connect() timed out', 'broken pipe')


def _rendezvous_retry(fn):
    """a 2-rank gloo rendezvous on a just-released port can lose a race with another process of the machine (seen once in a few hundred
    runs: a worker exits with 'address already in use'): one more attempt on a fresh port - but ONLY for such a rendezvous failure.
    An AssertionError (or any other exception) is the test's verdict and propagates at once: a nondeterministic ordering or race bug
    in the collectives must not be able to pass on its second try (round-5 advisor finding)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            return fn(*a, **kw)
        except AssertionError:
            raise
        except Exception as first:
            if not any(m in str(first).lower() for m in _RENDEZVOUS_MARKERS):
                raise
            return fn(*a, **kw)
    return wrapped


def _by_value(o):
    """tensors cross the result queue BY VALUE (numpy): a torch tensor is pickled as a file descriptor that the parent fetches from the
    worker's resource-sharer socket - gone if the worker has already exited (FileNotFoundError, seen when the machine is busy)"""
    if isinstance(o, torch.Tensor):
        return ('__tensor__', o.detach().cpu().numpy())
    if isinstance(o, (list, tuple)):
        return type(o)(_by_value(v) for v in o)
    if isinstance(o, dict):
        return {k: _by_value(v) for k, v in o.items()}
    return o


def _from_value(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == '__tensor__':
        return torch.from_numpy(o[1].copy())
    if isinstance(o, (list, tuple)):
        return type(o)(_from_value(v) for v in o)
    if isinstance(o, dict):
        return {k: _from_value(v) for k, v in o.items()}
    return o


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
    q.put(_by_value((rank, out, info)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_and_seeds():
    from conditional_score_diffusion_amd import distributed as D
    assert [D.shard_bounds(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    # a batch that does not divide (BASELINE configs[3]: 50 images over 8 GPUs): full shards first, a ragged / empty tail
    b = [D.shard_bounds(50, r, 8) for r in range(8)]
    assert b == [(0, 7), (7, 14), (14, 21), (21, 28), (28, 35), (35, 42), (42, 49), (49, 50)]
    assert [D.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [D.shard_bounds(3, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]      # the last rank is empty
    assert sum(hi - lo for lo, hi in b) == 50 and D.shard_size(50, 8) == 7
    assert len({D.rank_seed(42, r) for r in range(8)}) == 8


@_rendezvous_retry
def test_two_rank_sharded_sampling_gloo():
    from conditional_score_diffusion_amd import distributed as D
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=900)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    y = torch.arange(8 * 3 * 2 * 2, dtype=torch.float32).reshape(8, 3, 2, 2)
    expect = torch.cat([y[:4] * 2 + float(D.rank_seed(5, 0) % 97), y[4:] * 2 + float(D.rank_seed(5, 1) % 97)])
    for rank, out, info in res:
        assert torch.equal(out, expect)            # every rank holds the whole batch, shards in rank order
        assert info['n'] == 4 and info['seed'] == D.rank_seed(5, rank)


def _nonfinite_sampler(model, y, seed=0, **kw):
    from conditional_score_diffusion_amd._lib import NonFiniteError
    if dist.get_rank() == 1:
        raise NonFiniteError('stand-in: the state of this shard left the finite range')
    return y * 2.0, {'n': y.shape[0]}


def _nonfinite_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from conditional_score_diffusion_amd import distributed as D
    from conditional_score_diffusion_amd._lib import NonFiniteError
    y = torch.arange(8 * 3 * 2 * 2, dtype=torch.float32).reshape(8, 3, 2, 2)
    try:
        D.sample_sharded(_nonfinite_sampler, None, y_global=y, seed=5)
        verdict = 'returned'
    except NonFiniteError as e:
        verdict = 'NonFiniteError: ' + str(e)
    # both ranks are still in step: a healthy sampling call right after the failed one gathers normally
    out, _ = D.sample_sharded(_fake_sampler, None, y_global=y, seed=5)
    q.put(_by_value((rank, verdict, out)))
    dist.barrier()
    dist.destroy_process_group()


@_rendezvous_retry
def test_two_rank_non_finite_shard_raises_on_every_rank_gloo():
    """one rank's shard overflows (NonFiniteError of csd_pc_sample's finiteness contract): the failure crosses the group in front of
    the gather, EVERY rank raises, nobody is left waiting in the collective (round-5 advisor finding)"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nonfinite_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=600)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res[0][1].startswith('NonFiniteError') and 'another rank' in res[0][1]
    assert res[1][1].startswith('NonFiniteError') and 'stand-in' in res[1][1]
    assert torch.equal(res[0][2], res[1][2]) and res[0][2].shape[0] == 8


def _ragged_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conditional_score_diffusion_amd import distributed as D
    res = {}
    for n in (5, 1):        # 5 images over 2 ranks -> 3 + 2 (padded gather, sliced back); 1 image -> rank 1 is empty
        y = torch.arange(n * 3 * 2 * 2, dtype=torch.float32).reshape(n, 3, 2, 2)
        out, info = D.sample_sharded(_fake_sampler, None, y_global=y, seed=5, pad_shape=(3, 2, 2))
        res[n] = (out, info)
    # ragged data-parallel loss weights: the summed gradient is the global-batch mean
    w = torch.nn.Parameter(torch.zeros(1))
    from conditional_score_diffusion_amd.distributed import GradSync

    class _Flat:      # the two attributes GradSync reads
        params, offsets = [w], [0, 1]
    flat = _Flat()
    flat.grad = torch.zeros(1)
    w.grad = flat.grad
    sync = GradSync(flat)
    n_local = 3 if rank == 0 else 2
    x = torch.arange(5.0)[:3] if rank == 0 else torch.arange(5.0)[3:]
    sync.scale_loss((w * x).mean(), local_n=n_local, global_n=5).backward()
    sync.finish()
    res['grad'] = flat.grad.clone()
    q.put(_by_value((rank, res)))
    dist.barrier()
    dist.destroy_process_group()


@_rendezvous_retry
def test_two_rank_ragged_shards_gloo():
    """a global batch that does not divide by the world size: padded all_gather, sliced result, weighted loss"""
    from conditional_score_diffusion_amd import distributed as D
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=900)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    y5 = torch.arange(5 * 3 * 2 * 2, dtype=torch.float32).reshape(5, 3, 2, 2)
    expect5 = torch.cat([y5[:3] * 2 + float(D.rank_seed(5, 0) % 97), y5[3:] * 2 + float(D.rank_seed(5, 1) % 97)])
    y1 = torch.arange(1 * 3 * 2 * 2, dtype=torch.float32).reshape(1, 3, 2, 2)
    for rank, r in res:
        assert torch.equal(r[5][0], expect5) and r[5][1].get('n', 0) == (3 if rank == 0 else 2)
        assert torch.equal(r[1][0], y1 * 2 + float(D.rank_seed(5, 0) % 97))
        assert torch.allclose(r['grad'], torch.tensor([2.0]))         # mean of 0..4


def test_single_process_is_passthrough():
    from conditional_score_diffusion_amd import distributed as D
    y = torch.ones(4, 3, 2, 2)
    out, info = D.sample_sharded(_fake_sampler, None, y_global=y, seed=1)
    assert out.shape == y.shape and info['n'] == 4


# ---- global-norm exactness mode: the Langevin step size uses batch means over the GLOBAL batch ----
def _install_cpu_standins():
    """Test doubles for the two HIP ops the corrector calls (the kernels themselves are covered by -m gpu tests);
    what is under test here is the cross-rank reduction and the step-size algebra."""
    from conditional_score_diffusion_amd import ops

    def row_norms(x):
        return torch.norm(x.reshape(x.shape[0], -1), dim=-1)

    def affine_noise_step(x, score, z, p, a, c):
        xm = p * x + a * score
        return xm + c * z, xm

    ops.row_norms, ops.affine_noise_step = row_norms, affine_noise_step


def _score(x, t):
    return -(x - 0.3) * (1.0 + t.reshape(-1, 1, 1, 1))


def _global_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _install_cpu_standins()
    from conditional_score_diffusion_amd import sde_lib
    from conditional_score_diffusion_amd.sampling import correctors
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 3, 4, 4, generator=g)
    z = torch.randn(2, 6, 3, 4, 4, generator=g)
    lo, hi = rank * 3, rank * 3 + 3
    it = iter([z[0, lo:hi], z[1, lo:hi]])
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: next(it)
    try:
        corr = correctors.get_corrector('langevin_global')(sde_lib.VESDE(0.01, 50., 100), _score, 0.16, 2)
        out, out_mean = corr.update_fn(x[lo:hi], torch.full((3,), 0.5))
    finally:
        torch.randn_like = orig
    q.put(_by_value((rank, out, out_mean)))
    dist.barrier()
    dist.destroy_process_group()


@_rendezvous_retry
def test_global_norm_langevin_equals_single_process_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_global_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=900)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got, got_mean = torch.cat([r[1] for r in res]), torch.cat([r[2] for r in res])
    # the reference's Langevin corrector (sampling/correctors.py:88-108, alpha = 1) on the whole batch in ONE process
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 3, 4, 4, generator=g)
    z = torch.randn(2, 6, 3, 4, 4, generator=g)
    t = torch.full((6,), 0.5)
    for i in range(2):
        grad = _score(x, t)
        gn = torch.norm(grad.reshape(6, -1), dim=-1).mean()
        nn_ = torch.norm(z[i].reshape(6, -1), dim=-1).mean()
        step = (0.16 * nn_ / gn) ** 2 * 2
        x_mean = x + step * grad
        x = x_mean + torch.sqrt(step * 2) * z[i]
    assert torch.allclose(got, x, rtol=1e-5, atol=1e-6) and torch.allclose(got_mean, x_mean, rtol=1e-5, atol=1e-6)


# ---- data-parallel training: bucketed gradient all-reduce over the flat gradient buffer ----------------------------------
def _tiny_net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 3))


def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conditional_score_diffusion_amd import distributed as D, optim
    net = _tiny_net()
    flat = optim.FlatParams(net.parameters())
    sync = D.GradSync(flat, bucket_bytes=4 * 60)           # four buckets (the net has 435 parameters)
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    lo, hi = D.shard_bounds(8, rank, world)
    out = []
    for it in range(2):                                    # two steps: the hook bookkeeping resets
        flat.zero_grad()
        loss = (net(data[lo:hi] + it) ** 2).mean()
        sync.scale_loss(loss).backward()
        sync.finish()
        out.append(flat.grad.clone())
    q.put(_by_value((rank, out, len(sync.buckets))))
    dist.barrier()
    dist.destroy_process_group()


@_rendezvous_retry
def test_two_rank_bucketed_gradient_allreduce_gloo():
    from conditional_score_diffusion_amd import optim
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=900)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res[0][2] >= 3
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    for it in range(2):
        net = _tiny_net()
        flat = optim.FlatParams(net.parameters())
        (net(data + it) ** 2).mean().backward()            # one process, global batch
        for r in range(world):
            assert torch.allclose(res[r][1][it], flat.grad, rtol=1e-5, atol=1e-7)
        assert torch.equal(res[0][1][it], res[1][1][it])   # every rank holds the same reduced gradient


def _skewed_grad_worker(rank, world, port, q):
    """as _grad_worker, with UNEQUAL backward durations: rank r sleeps r * 30 ms inside the backward of every layer, so that rank 0 has
    launched all its buckets (and sits in finish()) long before rank 1 launches its first; one parameter gets no gradient at all in
    the second step (its bucket is launched by finish(), not by a hook)"""
    import time
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conditional_score_diffusion_amd import distributed as D, optim
    net = _tiny_net()
    flat = optim.FlatParams(net.parameters())
    sync = D.GradSync(flat, bucket_bytes=4 * 60)
    order = []
    launch = sync._launch
    sync._launch = lambda b: (order.append(b), launch(b))[1]
    for m in net:
        if isinstance(m, torch.nn.Linear):
            m.register_full_backward_hook(lambda mod, gi, go: time.sleep(0.03 * rank))
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    lo, hi = D.shard_bounds(8, rank, world)
    out, orders = [], []
    for it in range(2):
        flat.zero_grad()
        h = net[:4](data[lo:hi] + it)
        loss = (net[4](h) ** 2).mean() if it == 0 else (h ** 2).mean()      # step 1: the last layer takes no gradient
        sync.scale_loss(loss).backward()
        sync.finish()
        out.append(flat.grad.clone())
        orders.append(list(order))
        del order[:]
    q.put(_by_value((rank, out, orders)))
    dist.barrier()
    dist.destroy_process_group()


@_rendezvous_retry
def test_two_rank_allreduce_with_unequal_backward_durations_gloo():
    """GradSync with ranks whose backward passes take different times: the buckets are launched in the SAME order
    on both ranks - the order the gradients become final, the collective's matching rule - whatever the skew, a bucket whose hooks
    never fire is launched by finish(), and both ranks end with the same global-batch mean"""
    from conditional_score_diffusion_amd import optim
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_skewed_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_value(q.get(timeout=900)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res[0][2] == res[1][2]                               # same launch order on both ranks, both steps
    assert res[0][2][0] == sorted(res[0][2][0], reverse=True) and len(res[0][2][0]) >= 3      # last layers' bucket first
    assert sorted(res[0][2][1]) == sorted(res[0][2][0])         # every bucket reduced in the step without a last-layer gradient too
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    for it in range(2):
        net = _tiny_net()
        flat = optim.FlatParams(net.parameters())
        h = net[:4](data + it)
        ((net[4](h) ** 2).mean() if it == 0 else (h ** 2).mean()).backward()
        for r in range(world):
            assert torch.allclose(res[r][1][it], flat.grad, rtol=1e-5, atol=1e-7)
        assert torch.equal(res[0][1][it], res[1][1][it])


@_rendezvous_retry
def test_bench_grouped_branch_two_ranks_gloo():
    """bench.py's N > 1 protocol exactly as the driver launches it (torch.distributed.run, one process per rank): process group,
    W untimed + K timed steps between barriers, the ONE all_gather of the finished samples, MAX of the ranks' times, rank 0's line -
    on the CPU with gloo and a stand-in for the HIP sampler (`--stub-sampler`: rank r sleeps 2 (r + 1) ms per step)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    K, W, B = 6, 2, 3
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', str(K),
                        '--warmup', str(W), '--batch', str(B), '--stub-sampler'], capture_output=True, text=True, timeout=600, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]      # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == K and j['warmup'] == W and j['scaling'] == 'weak'
    assert j['config']['global_batch'] == 2 * B
    # the timed region is the slowest rank's (rank 1: 4 ms per step), not rank 0's 2 ms
    assert 3.9 <= j['ms_per_step'] < 40.0
    assert abs(j['value'] - 2 * B / (1000.0 * j['ms_per_step'] * 1e-3)) < 1e-9 * max(1.0, j['value'])
    # per-rank diagnostics (never part of `value`): rank r's own 2 (r + 1) ms per step; rank 0 waits in the all-gather for rank 1
    pr = j['per_rank']
    assert len(pr['ms_per_step']) == 2 and 1.9 <= pr['ms_per_step'][0] < pr['ms_per_step'][1] and pr['ms_per_step'][1] >= 3.9
    assert pr['all_gather_ms'][0] > pr['all_gather_ms'][1] and pr['all_gather_ms'][0] >= 0.5 * K * 2.0
    # the gather placed rank r's samples in block r: x started at r and every step added 1
    assert j['gathered_first_element_per_rank'] == [0.0 + K + W, 1.0 + K + W]
