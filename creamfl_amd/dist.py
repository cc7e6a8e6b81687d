"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) on the
xGMI mesh.  The reference is strictly single-GPU and time-multiplexes clients on cuda:0
(src/algorithms/MMFL.py:226-247); everything here is the build's own design:

  1. clients          -- `shard_clients`: the independent `for trainer in cur_trainers` bodies run on the rank that
                         OWNS the client (stable: client_idx % W, so a client's model persists across rounds);
                         `allgather_client_reps` all-gathers each rank's [M, D] public-set representations following
                         a host-known plan (no flags, no host sync, no zero blocks for uni-modal clients).
  2. con_w            -- `conw_aggregate_sharded`: the M rows of the log-prob are independent, so every rank
                         computes rows [r0, r1) for all C clients against the full (resident) global bank,
                         combines them locally and all-gathers its [M/W, D] block of the aggregate.
  3. global contrast  -- `DataParallelContext`: each rank encodes its own batch, features are all-gathered with
                         a gradient-aware gather, every rank evaluates the (cheap) full-batch pair loss, and
                         encoder gradients are averaged by `GradBuckets` (the build's own bucketed all-reduce: one
                         multi-tensor pack per bucket, deferred weight gradients included) overlapped with backward.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the representation all-gathers are <= 51 MB per rank
(latency/link bound, a few ms); the only bandwidth-significant collective is the encoder-gradient all-reduce
(~620 MB fp32 / 310 MB with bf16 trunk weights for ResNet-101 + BERT-base), hence buckets of tens of MB (32 MB default) overlapped
with backward.

The collectives take the compute kernels as arguments (defaults = the HIP ops) so that the sharding logic can
be exercised by world_size-2 gloo tests on CPU with the oracle injected (tests/test_dist_gloo.py); the
product never falls back to CPU on its own.
"""
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_gather_cat(t, group=None):
    """[n, ...] on every rank -> [W*n, ...] (rank-major).  No autograd."""
    rank, world = _world(group)
    if world == 1:
        return t
    t = t.contiguous()
    if dist.get_backend(group) == 'gloo':
        # test-only transport (CPU tests; two processes on one GPU in tests/test_gpu_multirank.py): gloo has no device
        # all-gather, so device tensors take a host round trip
        src = t.cpu() if t.is_cuda else t
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src, group=group)
        return torch.cat(parts, 0).to(t.device)
    out = t.new_empty((world * t.shape[0],) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def broadcast_module(module, src=0, group=None):
    """Make every rank's copy of `module` (parameters and buffers) identical to rank `src`'s."""
    rank, world = _world(group)
    if world == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


def average_buffers(module, group=None):
    """Replace every floating-point buffer of `module` (BatchNorm running statistics) by its mean over the ranks: ONE flat
    all-reduce.  Integer buffers (batch counters) are left alone -- they advance identically on every rank."""
    rank, world = _world(group)
    if world == 1:
        return
    with torch.no_grad():
        bufs = [b for b in module.buffers() if b.is_floating_point()]
        if not bufs:
            return
        flat = torch.cat([b.detach().float().reshape(-1) for b in bufs])
        if dist.get_backend(group) == 'gloo' and flat.is_cuda:
            host = flat.cpu()
            dist.all_reduce(host, group=group)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, group=group)
        flat /= world
        off = 0
        for b in bufs:
            n = b.numel()
            b.copy_(flat[off:off + n].view_as(b))
            off += n


def reseed_from_rank0(device=None, group=None):
    """Give every rank the same torch generator state (host and device): rank 0 draws a seed, all ranks seed with it."""
    rank, world = _world(group)
    if world == 1:
        return
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
    if dist.get_backend(group) != 'gloo' and device is not None:
        seed = seed.to(device)
    dist.broadcast(seed, 0, group=group)
    torch.manual_seed(int(seed.item()))


class _GatherWithGrad(torch.autograd.Function):
    """forward: all-gather rows; backward: this rank's slice of the upstream gradient times W.
    Every rank evaluates the SAME full-batch loss L on the gathered features, so dL/d(local rows) is the local
    slice of the full gradient; the factor W cancels the 1/W of DDP's gradient averaging, i.e. parameter
    gradients come out as the exact gradient of L (the single-process large-batch semantics)."""

    @staticmethod
    def forward(ctx, t, group):
        ctx.group = group
        ctx.rank, ctx.world = _world(group)
        ctx.n = t.shape[0]
        return all_gather_cat(t, group)

    @staticmethod
    def backward(ctx, g):
        sl = g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n]
        return sl * float(ctx.world), None


def gather_with_grad(t, group=None):
    return _GatherWithGrad.apply(t, group)


class GradBuckets:
    """Bucketed gradient all-reduce for the encoder replicas, overlapped with the backward pass -- the build's own
    replacement for torch DDP's reducer, which does not fit this step:
      * DDP copies every fresh gradient into its bucket with one copy kernel per PARAMETER (523 launches per step for
        ResNet-101 + BERT-base once `zero_grad(set_to_none=True)` has dropped the bucket views); here a bucket is packed by
        ONE multi-tensor copy when its last gradient has arrived;
      * DDP only sees gradients that autograd accumulates, so the convolution weight gradients could not stay deferred on
        the auxiliary stream (streams.py); here the deferred task reports its gradient itself (`notify`).
    Buckets are filled in reverse parameter order (the order gradients become ready), one open bucket per dtype,
    <= bucket_cap_mb each, and reduced strictly in completion order on every rank.  The pack + all-reduce run on a communication stream that
    first waits for every stream a gradient may have been produced on; `finish()` makes the caller's stream wait for it.
    After `finish()` every `p.grad` is a view into its (averaged) bucket: the optimizer reads the reduced values in place.
    RCCL: ring all-reduce over xGMI is per-link bound (large messages), but the LAST bucket's all-reduce cannot overlap anything:
    32 MB buckets (default; `bucket_cap_mb`, bench.py --bucket-mb, MMFL --bucket_mb) give ~10 buckets for the 310 MB of bf16 / fp32
    gradients of ResNet-101 + BERT-base -- >= 8 of them overlap the backward pass, the exposed tail is ~1/10 of the bytes."""

    def __init__(self, params, group=None, bucket_cap_mb=32, assign_grads=True):
        self.group = group
        self.assign_grads = assign_grads       # False: the optimizer reads the bucket views itself (grad_views)
        self._keep = []
        self.rank, self.world = _world(group)
        params = [p for p in params if p.requires_grad]
        # One open bucket per dtype (bf16 trunk weights and fp32 norm / head parameters alternate layer by layer: splitting
        # on every dtype change would give hundreds of one-parameter buckets); a bucket closes when the next parameter
        # would overflow it.  Buckets are ordered by the position of their LAST parameter = the moment they complete.
        cap = int(bucket_cap_mb * (1 << 20))
        open_b, closed = {}, []
        for pos, p in enumerate(reversed(params)):
            nb = p.numel() * p.element_size()
            cur = open_b.get(p.dtype)
            if cur is not None and cur['bytes'] + nb > cap:
                closed.append(cur)
                cur = None
            if cur is None:
                cur = open_b[p.dtype] = {'params': [], 'bytes': 0, 'last': pos}
            cur['params'].append(p)
            cur['bytes'] += nb
            cur['last'] = pos
        closed += list(open_b.values())
        closed.sort(key=lambda b: b['last'])
        self.buckets = [b['params'] for b in closed]
        self.bucket_of, self.flat, self.views, self.slices = {}, [], [], []
        for bi, plist in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
            views, slices, off = [], [], 0
            for p in plist:
                slices.append(flat[off:off + p.numel()])                       # the slot as a 1-D run of memory
                views.append(self._physical_view(slices[-1], p))              # the same bytes in p's shape and strides
                off += p.numel()
                self.bucket_of[p] = bi
            self.flat.append(flat)
            self.views.append(views)
            self.slices.append(slices)
        self._hooks = [p.register_post_accumulate_grad_hook(self.notify) for p in params]
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._next = 0
        self._main = None
        self.comm = None
        # 'idle' -> prepare() -> 'open' -> finish() -> 'reduced' -> consume() -> 'idle'.  A backward pass that was not bracketed
        # by prepare()/finish() leaves the bucket views holding the PREVIOUS step's averages: consume() refuses them.
        self.state = 'idle'
        self._no_grad = []                    # parameters that received no gradient in the current backward pass
        if params and params[0].is_cuda:
            from . import streams
            self.comm = streams.get(params[0].device, 'comm')
            streams.GRAD_READY[0] = self.notify          # deferred weight gradients report here (ops._ConvSplitFn)

    def close(self):
        """Detach this reducer from the model: remove the autograd hooks and the deferred-weight-gradient callback (a second
        reducer on the same parameters would otherwise pack and all-reduce every bucket twice per step)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.comm is not None:
            from . import streams
            if streams.GRAD_READY[0] == self.notify:
                streams.GRAD_READY[0] = None
        self.state = 'closed'

    @staticmethod
    def _physical_view(flat, p):
        """A view of `flat` with p's shape AND strides (channels_last convolution weights keep their memory order)."""
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            n, c, h, w = p.shape
            return flat.view(n, h, w, c).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def prepare(self):
        """Before every backward pass."""
        if self.state == 'closed':
            raise RuntimeError('GradBuckets.prepare() on a closed reducer')
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._next = 0
        self._keep = []
        self._no_grad = []
        self.state = 'open'
        p0 = self.buckets[0][0] if self.buckets else None
        self._main = torch.cuda.current_stream(p0.device) if (p0 is not None and p0.is_cuda) else None

    def notify(self, p):
        """A parameter's gradient for this step is complete (autograd hook, or a deferred weight-gradient task)."""
        bi = self.bucket_of.get(p)
        if bi is None or p in self._seen or self.state != 'open':
            return
        self._seen.add(p)
        self._ready[bi] += 1
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._reduce(self._next)
            self._next += 1

    def _reduce(self, bi):
        plist, views, flat = self.buckets[bi], self.views[bi], self.flat[bi]
        grads = [p.grad for p in plist]
        if self.comm is not None:
            from . import streams
            for s in [self._main] + streams.existing(flat.device):
                if s is not None and s != self.comm:
                    self.comm.wait_stream(s)                 # everything that produced these gradients
            ctx = torch.cuda.stream(self.comm)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx, torch.no_grad():
            # Pack: slot views carry the parameter's own strides, which are also the gradient's, so the whole bucket is ONE
            # multi-tensor copy.  Host work per parameter is kept to a list lookup: the backward pass of this step is close
            # to host-bound, and a Python reducer that touches every gradient object (as_strided / record_stream /
            # grad assignment: ~12 us each) measured +6.6 ms per step at 523 parameters.
            if all(g is not None for g in grads):
                fast = [g.stride() == v.stride() for g, v in zip(grads[:1], views[:1])][0]
                try:
                    torch._foreach_copy_(views, grads)
                except RuntimeError:
                    fast = False
                if not fast:
                    for v, g in zip(views, grads):
                        v.copy_(g)
            else:
                for v, g in zip(views, grads):
                    if g is None:
                        v.zero_()
                    elif g.data_ptr() != v.data_ptr():
                        v.copy_(g)
            if self.world > 1:
                if dist.get_backend(self.group) == 'gloo':
                    dist.all_reduce(flat, group=self.group)
                    flat.div_(self.world)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            # the local gradients were produced on other streams and are read here on the communication stream: keep them
            # alive until finish() has ordered the caller's stream behind it (cheaper than record_stream per tensor)
            self._keep.append(grads)
            # A parameter without a gradient in this backward pass (the same ones on every rank: e.g. a tower the KD phase
            # does not reach) keeps grad None: optimizers skip it -- no weight decay, no moment decay -- exactly as in the
            # single-process run.  Its (zero) slot still travels with the bucket.
            self._no_grad += [p for p, g in zip(plist, grads) if g is None]
            if self.assign_grads:
                for p, v, g in zip(plist, views, grads):
                    if g is not None:
                        p.grad = v

    def finish(self):
        """After the backward pass: reduce whatever is left (buckets holding parameters that got no gradient this step; the
        same ones on every rank), then make the caller's stream wait for the communication stream."""
        if self.state != 'open':
            raise RuntimeError('GradBuckets.finish() without prepare() before the backward pass')
        while self._next < len(self.buckets):
            if self._ready[self._next] > 0:
                self._reduce(self._next)
            else:
                self._no_grad += list(self.buckets[self._next])
            self._next += 1
        if self.comm is not None:
            torch.cuda.current_stream(self.flat[0].device).wait_stream(self.comm)
        self._keep = []
        self.state = 'reduced'

    def consume(self):
        """Called by the consumer of the bucket views (the fused optimizer) right before it reads them: the views are only
        valid for a backward pass that ran between prepare() and finish().  Returns the parameters that got NO gradient in
        that pass (their views hold zeros and must be skipped)."""
        if self.state != 'reduced':
            raise RuntimeError('gradient buckets are %s, not reduced: this backward pass was not bracketed by '
                               'prepare_backward() / finish_backward() and the bucket views hold stale gradients' % self.state)
        self.state = 'idle'
        return self._no_grad

    def comm_stats(self):
        """What one step moves through the gradient all-reduce: bytes (sum of the bucket sizes) and the bucket sizes in reduce
        order."""
        sizes = [f.numel() * f.element_size() for f in self.flat]
        return {'allreduce_bytes_per_step': int(sum(sizes)), 'buckets': len(sizes), 'bucket_bytes': sizes}

    def grad_views(self):
        """{parameter: its averaged gradient (bucket view)} -- valid after finish(); constant objects across steps."""
        return {p: v for plist, views in zip(self.buckets, self.views) for p, v in zip(plist, views)}


class DataParallelContext:
    """Large-batch global contrast across ranks: each rank encodes its own batch, the features
    are all-gathered with a gradient-aware gather, every rank evaluates the (cheap) full-batch pair loss, and the encoder
    gradients are averaged by GradBuckets while the backward pass is still running."""

    def __init__(self, model, group=None, bucket_cap_mb=32, assign_grads=True):
        self.group = group
        self.rank, self.world = _world(group)
        self.module = model
        broadcast_module(model, 0, group)
        self.reducer = GradBuckets(list(model.parameters()), group, bucket_cap_mb, assign_grads=assign_grads)

    def close(self):
        self.reducer.close()

    def gather_features(self, image_features, caption_features):
        return (gather_with_grad(image_features.float(), self.group),
                gather_with_grad(caption_features.float(), self.group))

    def prepare_backward(self):
        self.reducer.prepare()

    def finish_backward(self, criterion_params=None):
        """The criterion's scalars (shift, negative_scale) see the identical full-batch loss on every rank, so
        their gradients are already equal across ranks; only the encoder buckets are reduced."""
        self.reducer.finish()


# ------------------------------------------------------------------------------------- clients per GPU
def client_owner(trainer, position, world):
    """Rank that owns a client.  Keyed by the client's STABLE identity (`client_idx`, MMFL.py:176-178), not by its
    position in this round's `random.sample`: a client's model, optimizer state and epoch counter live only in the
    process that trains it and must be the ones it continues from in later rounds (MMFL.py:226-247 trains persistent
    trainer objects).  Objects without a `client_idx` fall back to their position."""
    key = getattr(trainer, 'client_idx', None)
    if key is None and isinstance(trainer, dict):
        key = trainer.get('client_idx')
    return (int(key) if key is not None else position) % world


def client_modalities(trainer):
    """('img',) | ('txt',) | ('img', 'txt'): which representations `generate_logits` of this client returns.  Known on
    the host of every rank (it is a property of the client's type), so nothing about it has to be communicated."""
    m = getattr(trainer, 'modalities', None)
    if m is None and isinstance(trainer, dict):
        m = trainer.get('modalities', tuple(k for k in ('img', 'txt') if trainer.get(k) is not None))
    if m is None:
        raise ValueError(f'cannot tell the modalities of client {trainer!r}')
    return tuple(m)


def client_plan(trainers, world):
    """plan[r] = [(position in `trainers`, modalities), ...] of the clients rank r owns this round, for EVERY rank
    (identical on all ranks: they sample the same client list)."""
    plan = [[] for _ in range(world)]
    for pos, t in enumerate(trainers):
        plan[client_owner(t, pos, world)].append((pos, client_modalities(t)))
    return plan


def shard_clients(trainers, rank=None, world=None, group=None):
    """This round's clients owned by this rank (stable ownership: client_owner)."""
    if rank is None or world is None:
        rank, world = _world(group)
    return [t for pos, t in enumerate(trainers) if client_owner(t, pos, world) == rank]


def balanced_sample(trainers, k, world=None, rng=None):
    """A round's clients chosen like `random.sample(trainers, k)` (MMFL.py:223) but spread over the ranks: clients are drawn in
    random order and one is taken only while its owner (client_owner: client_idx % world, the stable home of its model and
    optimizer state) has fewer than ceil(k / world) clients this round -- with 8 clients per round on 8 GPUs every rank trains
    exactly one.  Falls back to plain draws if the ownership map cannot be balanced.  Deterministic given `rng` (default: the
    module-level `random`, which MMFL's driver seeds identically on every rank)."""
    import random as _random
    rng = rng or _random
    if world is None:
        world = _world()[1]
    order = list(range(len(trainers)))
    rng.shuffle(order)
    cap = -(-k // max(1, world))
    load, picked, rest = {}, [], []
    for pos in order:
        r = client_owner(trainers[pos], pos, world)
        if len(picked) < k and load.get(r, 0) < cap:
            load[r] = load.get(r, 0) + 1
            picked.append(pos)
        else:
            rest.append(pos)
    picked += rest[:k - len(picked)]
    return [trainers[p] for p in picked]


def allgather_client_reps(local_reps, plan, M, D, device, group=None):
    """All-gather of the public-set representations of this round's clients.
    local_reps: one {'img': [M, D] | None, 'txt': [M, D] | None} per client this rank trained, in plan[rank] order.
    plan: client_plan(...) -- who holds which modality is host knowledge, so there is no flag exchange and no
    device->host synchronisation.  Per slot (the k-th client of every rank) one all-gather moves the FIRST representation
    of each rank's client, whatever its modality, and a second one only if some rank's client in that slot is
    multimodal; a rank contributes a zero block only where it has nothing for that (slot, sub-slot).
    Returns (img_vecs, txt_vecs): lists of [M, D] tensors in the order of the sampled client list."""
    rank, world = _world(group)
    slots = max((len(p) for p in plan), default=0)
    got = {}
    for s in range(slots):
        nsub = max((len(p[s][1]) if s < len(p) else 0) for p in plan)
        for k in range(nsub):
            mine = None
            if s < len(plan[rank]) and k < len(plan[rank][s][1]):
                mine = local_reps[s][plan[rank][s][1][k]]
                if mine is None or tuple(mine.shape) != (M, D):
                    raise RuntimeError(f'client in slot {s} did not return a [{M}, {D}] {plan[rank][s][1][k]} representation')
                mine = mine.to(device=device, dtype=torch.float32)
            else:
                mine = torch.zeros(M, D, dtype=torch.float32, device=device)
            bufs = all_gather_cat(mine.reshape(1, M, D), group)              # [W, M, D]
            for r in range(world):
                if s < len(plan[r]) and k < len(plan[r][s][1]):
                    got[(plan[r][s][0], plan[r][s][1][k])] = bufs[r]
    img_vecs = [got[key] for key in sorted(got) if key[1] == 'img']
    txt_vecs = [got[key] for key in sorted(got) if key[1] == 'txt']
    return img_vecs, txt_vecs


class RepGatherBuffer:
    """The round's representation exchange as ONE collective.  The buffer is [W, K, M, D] with K = the
    largest number of [M, D] blocks any rank contributes this round (one per modality of every client it owns; host knowledge:
    `client_plan`).  Clients write their public-set representations straight into `slot(j)` views of this rank's slice
    (`generate_logits(..., out=...)`: no torch.cat, no staging copy), then `gather()` runs one in-place
    all_gather_into_tensor over RCCL.  `wire_dtype=torch.bfloat16` halves the bytes on xGMI (410 -> 205 MB for 8 clients at
    D = 256); `blocks()` hands the con_w kernels fp32 tensors either way.  The buffer is reused across rounds."""

    def __init__(self, plan, M, D, device, wire_dtype=torch.float32, group=None):
        self.plan, self.M, self.D, self.group = plan, M, D, group
        self.rank, self.world = _world(group)
        self.K = max((sum(len(mods) for _, mods in p) for p in plan), default=0)
        self.buf = torch.zeros(self.world, max(self.K, 1), M, D, dtype=wire_dtype, device=device)
        # (position in the sampled client list, modality) of block j of rank r
        self.index = [[(pos, k) for pos, mods in p for k in mods] for p in plan]

    def matches(self, plan, M, D, wire_dtype):
        return (self.M, self.D, self.buf.dtype) == (M, D, wire_dtype) and \
            max((sum(len(m) for _, m in p) for p in plan), default=0) <= self.buf.shape[1] and len(plan) == self.world

    def rebind(self, plan):
        self.plan = plan
        self.K = max((sum(len(mods) for _, mods in p) for p in plan), default=0)
        self.index = [[(pos, k) for pos, mods in p for k in mods] for p in plan]

    def out_views(self, local_slot):
        """{'img' | 'txt': [M, D] view} for the local_slot-th client this rank owns this round."""
        j0 = sum(len(mods) for _, mods in self.plan[self.rank][:local_slot])
        mods = self.plan[self.rank][local_slot][1]
        return {k: self.buf[self.rank, j0 + i] for i, k in enumerate(mods)}

    def gather(self):
        if self.world == 1:
            return
        mine = self.buf[self.rank]
        if dist.get_backend(self.group) == 'gloo':
            src = mine.cpu() if mine.is_cuda else mine.clone()
            if src.dtype == torch.bfloat16:                      # gloo has no bf16 all-gather: move the raw bytes
                src = src.contiguous().view(torch.uint8)
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    self.buf[r].copy_((p.view(torch.bfloat16) if self.buf.dtype == torch.bfloat16 else p).to(self.buf.device))
            return
        dist.all_gather_into_tensor(self.buf.view(-1), mine.reshape(-1), group=self.group)      # in place: mine = buf[rank]

    def blocks(self):
        """(img_vecs, txt_vecs): fp32 [M, D] tensors in the order of the sampled client list."""
        got = {}
        for r in range(self.world):
            for j, key in enumerate(self.index[r]):
                t = self.buf[r, j]
                got[key] = t if t.dtype == torch.float32 else t.float()
        return ([got[k] for k in sorted(got) if k[1] == 'img'], [got[k] for k in sorted(got) if k[1] == 'txt'])


def row_shard(M, rank, world, align=128):
    """Contiguous row range of rank `rank` when M rows are split over `world` ranks in multiples of `align`."""
    per = -(-M // world)
    per = -(-per // align) * align
    r0 = min(M, rank * per)
    r1 = min(M, r0 + per)
    return r0, r1


def conw_aggregate_sharded(vecs, global_other, group=None, logprob_fn=None, combine_fn=None):
    """Row-sharded con_w (MMFL.py:298-335): returns the full aggregate [M, D] on every rank.
    vecs: list of C [M, D] tensors (all clients, identical on every rank after allgather_client_reps)."""
    if logprob_fn is None or combine_fn is None:
        from . import ops
        logprob_fn = logprob_fn or ops.conw_logprob
        combine_fn = combine_fn or ops.conw_combine
    rank, world = _world(group)
    M, D = vecs[0].shape
    if world == 1:
        lp = torch.stack([logprob_fn(v, global_other, 0, M) for v in vecs], 0)
        return combine_fn(vecs, lp)
    r0, r1 = row_shard(M, rank, world)
    per = row_shard(M, 0, world)[1]
    block = torch.zeros(per, D, dtype=torch.float32, device=vecs[0].device)
    if r1 > r0:
        lp = torch.stack([logprob_fn(v, global_other, r0, r1 - r0) for v in vecs], 0)      # [C, rows]
        block[:r1 - r0] = combine_fn([v[r0:r1] for v in vecs], lp)
    full = all_gather_cat(block, group)                                                    # [W*per, D]
    return full[:M].contiguous()
