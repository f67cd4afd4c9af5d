"""Batch-sharded sampling across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no explicit collective on the sampling path: under Lightning-DDP ``trainer.test``
each rank samples its own batches and writes them to disk.  This module is the
replacement for that implicit sharding (run_lib.py:95-102 + PairedCallback.py:166-181): every
image's trajectory is independent, so the condition batch ``y`` is cut into contiguous equal
shards, each rank runs the fused PC loop on its shard with NO per-step communication, and ONE
``all_gather`` (direct one-hop over xGMI: 2.4 MB per rank at 8 x 160^2 images) collects the result.

Exactness: the Langevin step size uses batch-mean norms (sampling/correctors.py:102-104).
* "per-shard" mode (default): a sharded run equals the reference run independently per shard - which is exactly
  what Lightning-DDP testing does; no per-step communication at all.
* "global-norm" mode (``global_norm=True``): identical to ONE reference process holding the global batch - every
  PC step all-reduces two fp32 sums (8 bytes, RCCL) between the library's csd_pc_step_begin / csd_pc_step_end
  calls (sampling/fused.py); everything stays on the device, the host never waits.
"""
import torch
import torch.distributed as dist


def shard_size(n_total, world):
    """Rows of a full shard: ceil(n_total / world).  The last shards may be shorter (or empty) when the batch does not divide."""
    return -(-int(n_total) // int(world))


def shard_bounds(n_total, rank, world):
    """Contiguous shards of ``shard_size`` rows; a global batch that does not divide evenly (configs[3]: 50 images over 8 GPUs ->
    7,7,7,7,7,7,7,1) leaves the last shard(s) ragged or empty.  ``sample_sharded`` pads those to the common size for the gather."""
    per = shard_size(n_total, world)
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def rank_seed(seed, rank):
    """Distinct Philox key per rank so shards do not reuse noise."""
    return (int(seed) * 1000003 + 7919 * int(rank)) & 0x7FFFFFFFFFFFFFFF


def sample_sharded(sampler, model, y_global=None, n_total=None, seed=None, group=None, global_norm=False, pad_shape=None,
                   **sampler_kw):
    """Run ``sampler`` on this rank's shard and all-gather the samples.

    sampler : ``fn(model, y_shard, seed=..., **kw) -> (x, info)`` for conditional sampling (as returned
              by ``get_conditional_sampling_fn``) or ``fn(model, seed=..., **kw)`` when ``y_global`` is None
              (unconditional; the sampler's ``shape`` must already be the per-rank shape).
    pad_shape : (C, H, W) of a sample - only needed when a rank can end up with an EMPTY shard (fewer images than ranks).
    Returns (samples of the GLOBAL batch on every rank, info of the local shard).
    """
    grouped = dist.is_initialized()       # a process group exists: EVERY collective of the path runs through it, also at world size 1
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    if seed is None:        # a fresh base key per call from torch's generator (rank_seed keeps the ranks' streams distinct)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    n_global = y_global.shape[0] if y_global is not None else n_total
    if global_norm and grouped:
        if not n_global:
            raise ValueError('global-norm mode needs the global batch size (y_global or n_total)')
        lo_last, hi_last = shard_bounds(n_global, world - 1, world)
        if hi_last <= lo_last:
            raise ValueError('global-norm mode: %d images leave a rank of %d without work (it would miss the per-step all-reduce)'
                             % (n_global, world))
        sampler_kw['global_norm'] = (lambda sums: dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group), int(n_global))
    # A shard whose state leaves the finite range raises NonFiniteError on ITS rank only (the finiteness contract of csd_pc_sample is
    # per call).  Raised here, in front of the gather, it would leave the healthy ranks blocked in the collective until the RCCL
    # timeout: the failure is carried across the group first (one MAX all-reduce of a flag) and then raised on EVERY rank.
    from ._lib import NonFiniteError
    failure, x_local, info = None, None, {}
    try:
        if y_global is not None:
            lo, hi = shard_bounds(n_global, rank, world)
            if hi > lo:
                x_local, info = sampler(model, y_global[lo:hi].contiguous(), seed=rank_seed(seed, rank), **sampler_kw)
            # (else: an empty ragged shard - nothing to sample, an all-zero block for the gather)
        else:
            lo, hi = 0, None
            x_local, info = sampler(model, seed=rank_seed(seed, rank), **sampler_kw)
    except NonFiniteError as e:
        if not grouped or global_norm:      # (global-norm mode: every rank sees the same global norms and fails in the same step)
            raise
        failure = e
    if not grouped:
        return x_local, info
    if not global_norm:
        fdev = x_local.device if x_local is not None else (y_global.device if y_global is not None else torch.device('cpu'))
        if dist.get_backend(group) == 'nccl' and fdev.type != 'cuda':
            fdev = torch.device('cuda', torch.cuda.current_device())
        flag = torch.tensor([1.0 if failure is not None else 0.0], dtype=torch.float32, device=fdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if failure is not None:
            raise failure
        if float(flag.item()) > 0:
            raise NonFiniteError('sample_sharded: the shard of another rank left the finite range (its rank raises the details); '
                                 'csd_precision = \'fp32\' is the escape')
    if y_global is not None:
        per = shard_size(n_global, world)
        if x_local is None or x_local.shape[0] < per:           # pad the ragged shard to the common gather size
            shape_tail = tuple(x_local.shape[1:]) if x_local is not None else tuple(pad_shape or ())
            if x_local is None and not shape_tail:
                raise ValueError('an empty shard needs pad_shape=(C, H, W) to take part in the gather')
            # the padding block lives where the samples live (an empty shard: on the model's device), in their dtype - the gather needs
            # matching tensors on every rank, whatever device y_global happens to be on
            if x_local is not None:
                pad = x_local.new_zeros((per,) + shape_tail)
                pad[:x_local.shape[0]] = x_local
            else:
                pad = torch.zeros((per,) + shape_tail, dtype=torch.float32, device=getattr(model, 'device', y_global.device))
            x_local = pad
    x_local = x_local.contiguous()
    out = torch.empty((world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                      device=x_local.device)
    dist.all_gather_into_tensor(out, x_local, group=group)      # the single collective of the path
    if y_global is not None and out.shape[0] != n_global:       # drop the padding rows of the ragged tail
        out = out[:n_global]
    return out, info


# ------------------------------------------------------------------------------------------------------------------
# data-parallel training: gradient all-reduce (replaces Lightning-DDP / NCCL, run_lib.py:55-73)
# ------------------------------------------------------------------------------------------------------------------
def ctypes_ptr(v):
    import ctypes
    return ctypes.c_void_p(int(v))


def bucket_first_modules(param_names, bucket_param_indices):
    """Lowest ``all_modules`` index among the parameters of every bucket (``all_modules.<k>.<leaf>``: the reference's module list,
    models/ddpm.py:97-147) - the module whose backward completes the bucket, since the backward walks the list back to front.
    None if a name does not follow the pattern."""
    mods = []
    for n in param_names:
        parts = n.split('.')
        if len(parts) < 3 or parts[0] != 'all_modules' or not parts[1].isdigit():
            return None
        mods.append(int(parts[1]))
    return [min(mods[i] for i in idxs) for idxs in bucket_param_indices]


class GradSync:
    """Bucketed gradient all-reduce over a flat gradient buffer, overlapped with the backward pass.

    The gradients of all parameters live in ONE contiguous buffer (optim.FlatParams), cut into ``bucket_bytes`` slices at
    parameter boundaries.  A post-accumulate hook per parameter counts arrivals; when the last gradient of a bucket has been
    written, ONE asynchronous ``all_reduce(SUM)`` of that slice is enqueued (RCCL runs it on its own stream while the
    backward kernels of earlier layers continue).  The loss is scaled by 1/world before ``backward`` (``scale_loss``), so the
    summed gradients ARE the global-batch mean - no extra pass over the buffer.  ``finish()`` waits for the outstanding
    handles (and reduces buckets whose hooks did not all fire, e.g. parameters without gradient this step).

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s); a ring all-reduce over 8 GPUs moves 2*(7/8) of the bucket per
    link, so 32 MiB buckets take ~0.4 ms each - large enough to run at link bandwidth, small enough that the first bucket
    (the last layers' gradients) is in flight while >90 % of the backward is still ahead.
    """

    def __init__(self, flat, group=None, bucket_bytes=32 << 20):
        self.flat, self.group = flat, group
        self.grouped = dist.is_initialized()      # with a process group the reduction runs through it, also at world size 1
        self.world = dist.get_world_size(group) if self.grouped else 1
        per = max(1, bucket_bytes // 4)
        self.buckets = []           # (lo, hi, [param indices])
        lo, idxs = 0, []
        for i, p in enumerate(flat.params):
            idxs.append(i)
            hi = int(flat.offsets[i + 1])
            if hi - lo >= per or i == len(flat.params) - 1:
                self.buckets.append((lo, hi, idxs))
                lo, idxs = hi, []
        self._bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self._bucket_of[i] = b
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._closed = False
        self._hook_handles = []     # removed by close(): a GradSync that is replaced must not keep reducing the shared buffer
        if self.grouped:
            for i, p in enumerate(flat.params):
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def attach_planned(self, model):
        """Planned training graph (the whole backward is ONE csd_unet_backward call: no autograd hooks fire): register one
        gradient-ready event per bucket on the network handle - the library records it on the backward's stream as soon as the
        gradients of that bucket's modules are final (csd_unet_backward_marks) - so that ``finish()`` can launch every bucket's
        all-reduce on a communication stream that waits for ITS event only: the late layers' gradients are reduced while the
        backward of the early layers still runs.  Without a process group (or off the GPU) nothing is registered."""
        self._model, self._events, self._comm, self._epoch = None, None, None, 0
        if not self.grouped or not self.flat.grad.is_cuda or getattr(model, '_h', None) is None:
            return False
        from ._lib import check, lib
        # names of the FLAT buffer's parameters, in its order (a frozen parameter is not in it: indices into the model's full
        # parameter list would be shifted - round-4 advisor finding)
        name_of = {id(p): n for n, p in model.named_parameters()}
        if any(id(p) not in name_of for p in self.flat.params):
            return False
        first = bucket_first_modules([name_of[id(p)] for p in self.flat.params], [b[2] for b in self.buckets])
        if first is None:
            return False
        import ctypes
        self._events = [lib().csd_event_create() for _ in self.buckets]
        if not all(self._events):
            raise RuntimeError('csd_event_create failed')
        fm = (ctypes.c_int * len(first))(*first)
        ev = (ctypes.c_void_p * len(first))(*self._events)
        check(lib().csd_unet_backward_marks(model._h, fm, ev, len(first)), 'unet_backward_marks')
        self._model = model
        # the handle has ONE set of marks: whoever attached last owns it.  A GradSync that was replaced on the same model (a second
        # Trainer built before the first is finalized) finds another owner in detach_planned() and leaves the marks alone
        model._marks_owner = self
        self._comm = torch.cuda.Stream(device=self.flat.grad.device)
        self._epoch = int(lib().csd_unet_backward_marks_epoch(model._h))
        self.overlapped_launches = 0      # buckets launched from their event (statistics for tests / logs)
        return True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        """Retire this GradSync: its autograd hooks are removed (they would otherwise launch a second all-reduce of the shared flat
        gradient next to the successor's), its gradient-ready events are destroyed, and the network handle's marks are erased
        only if they are still THIS object's (round-5 advisor finding).  Idempotent."""
        self._closed = True
        for h in getattr(self, '_hook_handles', []):
            h.remove()
        self._hook_handles = []
        self.detach_planned()

    def detach_planned(self):
        if getattr(self, '_events', None):
            from ._lib import lib
            model = self._model
            if getattr(model, '_marks_owner', None) is self:
                lib().csd_unet_backward_marks(model._h, None, None, 0)
                model._marks_owner = None
            for e in self._events:
                lib().csd_event_destroy(e)
        self._events = None

    def scale_loss(self, loss, local_n=None, global_n=None):
        """The summed gradients must be the GLOBAL-batch mean: equal shards -> loss / world; ragged shards (a global batch that does
        not divide, ``shard_bounds``) -> the rank's mean weighted by its share ``local_n / global_n`` of the images."""
        if local_n is not None and global_n:
            return loss * (float(local_n) / float(global_n))
        return loss if self.world == 1 else loss / self.world

    def _make_hook(self, i):
        def hook(param):
            if self._closed:
                return
            if getattr(self, '_events', None):        # planned graph: nothing was accumulated (the hook fires for an undefined
                return                                # gradient too); finish() launches every bucket from its gradient-ready event
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self._launched[b] = True
        self._handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after ``backward``: every bucket reduced, the flat gradient holds the global mean."""
        if self.grouped:
            ev = getattr(self, '_events', None)
            if ev and not any(self._launched):
                from ._lib import check, lib
                epoch = int(lib().csd_unet_backward_marks_epoch(self._model._h))
                # exactly one planned backward since the last step recorded every event - AND it wrote the .grad views itself: when
                # it did not (a gradient buffer the library cannot write directly), autograd accumulates into .grad on the main stream
                # AFTER the events, and a communication stream that waits for the events alone would reduce stale gradients.  Then
                # the buckets are launched below, from the current stream (behind the accumulation).
                if epoch == self._epoch + 1 and getattr(self._model, '_last_backward_direct', False):
                    for b in reversed(range(len(self.buckets))):      # the order the gradients become final
                        check(lib().csd_stream_wait_event(ctypes_ptr(self._comm.cuda_stream), ev[b]), 'stream_wait_event')
                        with torch.cuda.stream(self._comm):
                            self._launch(b)
                        self.overlapped_launches += 1
                self._epoch = epoch
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
        self._handles = []
        self._pending = [len(b[2]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
