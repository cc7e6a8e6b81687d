"""world_size-2 gloo tests (CPU) of the multi-GPU layer in creamfl_amd/dist.py.  The collectives and the
sharding logic are the product's; the compute kernels are injected from the oracle because the HIP ops need
a GPU (the product itself never falls back to CPU)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from creamfl_amd import dist as cdist


def _init(rank, world, path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    dist.init_process_group('gloo', init_method=f'file://{path}', rank=rank, world_size=world)
    torch.set_num_threads(2)


def _unit(gen, *s):
    return torch.nn.functional.normalize(torch.randn(*s, generator=gen), dim=-1)


# ------------------------------------------------------------------------------ global contrast across ranks
def _worker_global_contrast(rank, world, path, out):
    _init(rank, world, path)
    torch.manual_seed(0)
    enc_i = torch.nn.Linear(12, 8)
    enc_t = torch.nn.Linear(10, 8)
    model = torch.nn.ModuleDict({'i': enc_i, 't': enc_t})
    gen = torch.Generator().manual_seed(1)
    X = torch.randn(world * 6, 12, generator=gen)
    Y = torch.randn(world * 6, 10, generator=gen)
    a = torch.tensor([5.0], requires_grad=True)
    b = torch.tensor([3.0], requires_grad=True)

    class Both(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, y):
            return (torch.nn.functional.normalize(self.m['i'](x), dim=-1),
                    torch.nn.functional.normalize(self.m['t'](y), dim=-1))

    wrapped = Both(model)
    dp = cdist.DataParallelContext(wrapped, bucket_cap_mb=0.0002)        # ~200 bytes per bucket: several buckets, in order
    xs, ys = X[rank * 6:(rank + 1) * 6], Y[rank * 6:(rank + 1) * 6]
    fi, ft = dp.module(xs, ys)
    gi, gt = dp.gather_features(fi, ft)
    loss, _ = oracle.pair_loss_literal(gi, gt, a, b)
    dp.prepare_backward()
    loss.backward()
    dp.finish_backward()
    # after finish() every gradient is a view into its (averaged) bucket, in the parameter's own layout
    ok_views = all(p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == p.stride()
                   for plist, views in zip(dp.reducer.buckets, dp.reducer.views) for p, v in zip(plist, views))
    grads = {n: p.grad.clone() for n, p in wrapped.named_parameters()}
    if rank == 0:
        # single-process large-batch reference
        torch.manual_seed(0)
        ri, rt = torch.nn.Linear(12, 8), torch.nn.Linear(10, 8)
        a2 = torch.tensor([5.0], requires_grad=True)
        b2 = torch.tensor([3.0], requires_grad=True)
        l2, _ = oracle.pair_loss_literal(torch.nn.functional.normalize(ri(X), dim=-1),
                                         torch.nn.functional.normalize(rt(Y), dim=-1), a2, b2)
        l2.backward()
        ok = abs(loss.item() - l2.item()) < 1e-4 * abs(l2.item()) and ok_views and len(dp.reducer.buckets) >= 1
        ok &= torch.allclose(grads['m.i.weight'], ri.weight.grad, rtol=1e-4, atol=1e-6)
        ok &= torch.allclose(grads['m.t.bias'], rt.bias.grad, rtol=1e-4, atol=1e-6)
        ok &= torch.allclose(a.grad, a2.grad, rtol=1e-4) and torch.allclose(b.grad, b2.grad, rtol=1e-4)
        out.put(bool(ok))
    dist.destroy_process_group()


# ------------------------------------------------------------------------------ client reps + sharded con_w
def _worker_clients_conw(rank, world, path, out):
    _init(rank, world, path)
    M, D = 300, 16
    gen = torch.Generator().manual_seed(7)
    G_img, G_txt = _unit(gen, M, D), _unit(gen, M, D)
    # 3 clients this round: img, txt, mm  -> rank 0 gets clients 0 and 2, rank 1 gets client 1 (no client_idx: positional)
    reps_all = [{'img': _unit(gen, M, D), 'txt': None}, {'img': None, 'txt': _unit(gen, M, D)},
                {'img': _unit(gen, M, D), 'txt': _unit(gen, M, D)}]
    mine = cdist.shard_clients(reps_all)
    plan = cdist.client_plan(reps_all, world)
    ok = plan == [[(0, ('img',)), (2, ('img', 'txt'))], [(1, ('txt',))]]
    img_vecs, txt_vecs = cdist.allgather_client_reps(mine, plan, M, D, torch.device('cpu'))
    ok &= len(img_vecs) == 2 and len(txt_vecs) == 2
    ok &= torch.equal(img_vecs[0], reps_all[0]['img']) and torch.equal(img_vecs[1], reps_all[2]['img'])
    ok &= torch.equal(txt_vecs[0], reps_all[1]['txt']) and torch.equal(txt_vecs[1], reps_all[2]['txt'])

    def lp_fn(v, g, r0, rows):
        return oracle.conw_logprob(v, g, literal=False)[r0:r0 + rows]

    def comb_fn(vs, lp):
        w = torch.softmax(lp, 0)
        return sum(w[c][:, None] * vs[c] for c in range(len(vs)))

    agg = cdist.conw_aggregate_sharded(img_vecs, G_txt, logprob_fn=lp_fn, combine_fn=comb_fn)
    want, _, _ = oracle.conw_aggregate(img_vecs, G_txt, literal=False)
    ok &= torch.allclose(agg, want, rtol=1e-5, atol=1e-6) and agg.shape == (M, D)
    # replica resynchronisation
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    cdist.broadcast_module(lin)
    ok &= bool(torch.all(lin.weight == 1.0))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()


# ------------------------------------------------------------------------------ BASELINE configs[2]: 8 clients / round
class _FakeClient:
    """Stands for a ClientTrainer / MMClientTrainer: a stable `client_idx`, host-known `modalities`, per-process training
    state that must persist on the owning rank, and a deterministic representation."""

    def __init__(self, client_idx, modalities, M, D):
        self.client_idx, self.modalities, self.M, self.D = client_idx, modalities, M, D
        self.rounds_trained = 0                              # lives only in the process that trains the client

    def run_and_generate(self, round_n):
        self.rounds_trained += 1
        out = {'img': None, 'txt': None}
        for j, k in enumerate(self.modalities):
            gen = torch.Generator().manual_seed(1000 * self.client_idx + 10 * self.rounds_trained + (0 if k == 'img' else 1))
            out[k] = _unit(gen, self.M, self.D)
        return out


def _worker_config2_eight_clients(rank, world, path, out):
    """10 image + 10 text + 5 multimodal clients, 8 sampled per round (BASELINE configs[2]), two rounds with different
    samples: ownership is stable (client_idx % W), every representation arrives exactly once in sampled order, and a
    client sampled in both rounds continues from ITS OWN state on its owner (rounds_trained == 2 there, 0 elsewhere)."""
    import random
    _init(rank, world, path)
    M, D = 64, 8
    pool = ([_FakeClient(i + 1, ('img',), M, D) for i in range(10)] + [_FakeClient(11 + i, ('txt',), M, D) for i in range(10)]
            + [_FakeClient(21 + i, ('img', 'txt'), M, D) for i in range(5)])
    ok = True
    rng = random.Random(5)                                   # every rank samples the same clients (MMFL.train)
    seen_twice = None
    times = {}                                               # how often each client has been sampled (host knowledge)
    for round_n in range(2):
        cur = rng.sample(pool, 8) if round_n == 0 else ([seen_twice] + rng.sample([c for c in pool if c is not seen_twice], 7))
        if round_n == 0:
            seen_twice = cur[3]
        mine = cdist.shard_clients(cur)
        ok &= all(c.client_idx % world == rank for c in mine)
        plan = cdist.client_plan(cur, world)
        ok &= sorted(pos for p in plan for pos, _ in p) == list(range(8))
        local = [c.run_and_generate(round_n) for c in mine]
        img_vecs, txt_vecs = cdist.allgather_client_reps(local, plan, M, D, torch.device('cpu'))
        want_img, want_txt = [], []
        for c in cur:                                        # what a single process would have collected, in sampled order
            trained = times[c.client_idx] = times.get(c.client_idx, 0) + 1
            for k in c.modalities:
                gen = torch.Generator().manual_seed(1000 * c.client_idx + 10 * trained + (0 if k == 'img' else 1))
                (want_img if k == 'img' else want_txt).append(_unit(gen, M, D))
        ok &= len(img_vecs) == len(want_img) and len(txt_vecs) == len(want_txt)
        ok &= all(torch.equal(a, b) for a, b in zip(img_vecs, want_img))
        ok &= all(torch.equal(a, b) for a, b in zip(txt_vecs, want_txt))
    owner = seen_twice.client_idx % world
    ok &= seen_twice.rounds_trained == (2 if rank == owner else 0)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()



# ------------------------------------------------------------------------------ reducer life cycle
def _worker_reducer_lifecycle(rank, world, path, out):
    """(1) a backward pass that is NOT bracketed by prepare/finish (the round-2 KD step) must not hand the optimizer the
    previous step's averages: consume() raises; (2) bracketed, a second backward (the KD step after a contrastive step)
    yields the mean of the per-rank gradients; (3) a parameter without a gradient keeps grad None (no zero gradient that
    would make an optimizer decay it); (4) close() detaches the hooks so that a second reducer does not double-reduce."""
    _init(rank, world, path)
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({'a': torch.nn.Linear(6, 5), 'b': torch.nn.Linear(5, 3), 'unused': torch.nn.Linear(4, 4)})
    dp = cdist.DataParallelContext(net, bucket_cap_mb=0.0001)
    red = dp.reducer
    gen = torch.Generator().manual_seed(10 + rank)
    x1, x2 = torch.randn(4, 6, generator=gen), torch.randn(4, 6, generator=gen)
    ok = True

    def loss_of(x):
        return net['b'](torch.tanh(net['a'](x))).pow(2).mean()

    # step 1, bracketed
    dp.prepare_backward()
    loss_of(x1).backward()
    dp.finish_backward()
    ok &= red.state == 'reduced'
    skipped = red.consume()
    ok &= {id(p) for p in skipped} == {id(p) for p in net['unused'].parameters()}
    ok &= all(p.grad is None for p in net['unused'].parameters())
    # step 2 WITHOUT the bracket: the views still hold step 1 -> refused
    for p in net.parameters():
        p.grad = None
    loss_of(x2).backward()
    try:
        red.consume()
        ok = False
    except RuntimeError:
        pass
    # step 2 bracketed: mean over ranks of the local gradients
    for p in net.parameters():
        p.grad = None
    dp.prepare_backward()
    loss_of(x2).backward()
    local = net['a'].weight.grad.clone() if red.world == 1 else None
    dp.finish_backward()
    red.consume()
    got = red.grad_views()[net['a'].weight].clone()
    # reference: every rank recomputes both ranks' local gradients
    want = torch.zeros_like(got)
    for r in range(world):
        g2 = torch.Generator().manual_seed(10 + r)
        torch.randn(4, 6, generator=g2)
        xr = torch.randn(4, 6, generator=g2)
        (gr,) = torch.autograd.grad(loss_of(xr), net['a'].weight)
        want += gr / world
    ok &= torch.allclose(got, want, rtol=1e-5, atol=1e-7)
    # close(): hooks gone; a fresh reducer on the same parameters works alone
    n_hooks = len(red._hooks)
    dp.close()
    ok &= n_hooks > 0 and red._hooks == [] and red.state == 'closed'
    dp2 = cdist.DataParallelContext(net, bucket_cap_mb=1)
    for p in net.parameters():
        p.grad = None
    dp2.prepare_backward()
    loss_of(x2).backward()
    dp2.finish_backward()
    ok &= red._seen == set() or red.state == 'closed'          # the closed reducer saw nothing of this pass
    ok &= torch.allclose(dp2.reducer.grad_views()[net['a'].weight], want, rtol=1e-5, atol=1e-7)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()


def _worker_adamp_broadcast_state(rank, world, path, out):
    """AdamP.broadcast_state: masters, both moments and the per-parameter step counts of rank 0 reach every rank in a few
    flat collectives, channels_last state keeps its layout (host logic: the state is fabricated, no step is taken)."""
    from creamfl_amd.algorithms.optimizers import AdamP
    _init(rank, world, path)
    torch.manual_seed(100 + rank)                                 # DIFFERENT values per rank before the broadcast
    conv = torch.nn.Conv2d(4, 6, 3).to(memory_format=torch.channels_last)
    lin = torch.nn.Linear(5, 7)
    scal = torch.nn.Parameter(torch.randn(1))
    opt = AdamP(list(conv.parameters()) + list(lin.parameters()) + [scal], lr=1e-3)
    for i, p in enumerate(opt.param_groups[0]['params']):
        st = opt.state[p]
        st['step'] = 3 + i + 10 * rank
        st['exp_avg'] = torch.randn_like(p, memory_format=torch.preserve_format)
        st['exp_avg_sq'] = torch.rand_like(p, memory_format=torch.preserve_format)
        if p.dim() == 4:
            st['master'] = torch.randn_like(p, memory_format=torch.preserve_format)
    # a captured step with replays the host has not folded into the counts yet (AdamP.CaptureHandle): broadcast_state folds them
    # in BEFORE the counts travel (rank 0's 3 + i + 2 reach everybody) and voids the capture (its offsets are differences of counts)
    opt.prepare_capture()
    handle = opt.capture_begin()
    handle.params.extend(opt.param_groups[0]['params'])
    opt.capture_end(handle)
    handle.replayed()
    handle.replayed()
    opt.broadcast_state(0)
    ok = True
    ok &= handle.pending == 0 and not handle.valid()
    for i, p in enumerate(opt.param_groups[0]['params']):
        opt.state[p]['step'] -= 2                                 # (the rest of the test checks the fabricated counts)
    torch.manual_seed(100)                                        # rank 0's draws, replayed
    conv0 = torch.nn.Conv2d(4, 6, 3).to(memory_format=torch.channels_last)
    lin0 = torch.nn.Linear(5, 7)
    scal0 = torch.nn.Parameter(torch.randn(1))
    for i, (p, q) in enumerate(zip(opt.param_groups[0]['params'], list(conv0.parameters()) + list(lin0.parameters()) + [scal0])):
        st = opt.state[p]
        ok &= st['step'] == 3 + i
        m = torch.randn_like(q, memory_format=torch.preserve_format)
        v = torch.rand_like(q, memory_format=torch.preserve_format)
        ok &= torch.equal(st['exp_avg'], m) and torch.equal(st['exp_avg_sq'], v)
        ok &= st['exp_avg'].stride() == p.stride()
        if p.dim() == 4:
            ok &= torch.equal(st['master'], torch.randn_like(q, memory_format=torch.preserve_format))
            ok &= st['master'].is_contiguous(memory_format=torch.channels_last)
    #  state held by rank 0 ONLY (it resumed from a checkpoint, the others did not): every rank still joins every
    # collective -- no early return that would leave rank 0 alone in a broadcast -- and ends up with rank 0's state; state that
    # rank 0 does not hold is dropped.
    torch.manual_seed(7)
    lin2 = torch.nn.Linear(3, 4)
    extra = torch.nn.Parameter(torch.randn(2))
    opt2 = AdamP(list(lin2.parameters()) + [extra], lr=1e-3)
    if rank == 0:
        for i, p in enumerate(lin2.parameters()):
            opt2.state[p].update(step=5 + i, exp_avg=torch.full_like(p, 0.25 + i), exp_avg_sq=torch.full_like(p, 2.0 + i))
    else:
        opt2.state[extra].update(step=9, exp_avg=torch.ones_like(extra), exp_avg_sq=torch.ones_like(extra))
    opt2.broadcast_state(0)
    for i, p in enumerate(lin2.parameters()):
        st = opt2.state[p]
        ok &= st['step'] == 5 + i and bool((st['exp_avg'] == 0.25 + i).all()) and bool((st['exp_avg_sq'] == 2.0 + i).all())
    ok &= not opt2.state.get(extra)
    AdamP([torch.nn.Parameter(torch.randn(3))], lr=1e-3).broadcast_state(0)       # nobody holds anything: returns, no hang
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()


# ------------------------------------------------------------------------------ (8f-3) one-collective representation exchange
def _worker_rep_gather_buffer(rank, world, path, out):
    """RepGatherBuffer: clients write their [M, D] representations into views of the rank's slice, ONE all-gather moves
    everything, blocks come back in sampled-client order -- fp32 wire bit-exact, bf16 wire = the bf16-rounded values, and the
    con_w weights computed from bf16-wire representations stay within 1e-3 of the fp32 ones."""
    _init(rank, world, path)
    M, D = 96, 16
    pool = [_FakeClient(1, ('img',), M, D), _FakeClient(2, ('txt',), M, D), _FakeClient(3, ('img', 'txt'), M, D),
            _FakeClient(4, ('img',), M, D), _FakeClient(6, ('img', 'txt'), M, D)]
    plan = cdist.client_plan(pool, world)
    ok = True
    gen = torch.Generator().manual_seed(3)
    G_txt = _unit(gen, M, D)
    weights = {}
    for wire in (torch.float32, torch.bfloat16):
        buf = cdist.RepGatherBuffer(plan, M, D, torch.device('cpu'), wire)
        ok &= buf.K == max(sum(len(m) for _, m in p) for p in plan)
        mine = cdist.shard_clients(pool)
        for c in pool:
            c.rounds_trained = 0
        for slot, c in enumerate(mine):
            views = buf.out_views(slot)
            ok &= tuple(views) == c.modalities
            rep = c.run_and_generate(0)
            for k in c.modalities:                              # what generate_logits(out=...) does, batch by batch
                views[k][:M // 2].copy_(rep[k][:M // 2])
                views[k][M // 2:].copy_(rep[k][M // 2:])
        buf.gather()
        img_vecs, txt_vecs = buf.blocks()
        want_img, want_txt = [], []
        for c in pool:
            for k in c.modalities:
                g2 = torch.Generator().manual_seed(1000 * c.client_idx + 10 + (0 if k == 'img' else 1))
                (want_img if k == 'img' else want_txt).append(_unit(g2, M, D))
        ok &= len(img_vecs) == len(want_img) == 4 and len(txt_vecs) == len(want_txt) == 3
        for got, want in zip(img_vecs + txt_vecs, want_img + want_txt):
            ok &= got.dtype == torch.float32
            ok &= torch.equal(got, want if wire == torch.float32 else want.to(torch.bfloat16).float())
        _, w, _ = oracle.conw_aggregate(img_vecs, G_txt, literal=False)
        weights[wire] = w
        # the buffer is reusable with another plan of the same geometry
        ok &= buf.matches(plan, M, D, wire) and not buf.matches(plan, M, D + 1, wire)
    ok &= bool((weights[torch.float32] - weights[torch.bfloat16]).abs().max() < 1e-3)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()


def _worker_average_buffers_and_comm_stats(rank, world, path, out):
    """ data-parallel server phases leave per-shard BatchNorm running statistics on every rank: average_buffers makes
    them the mean over the ranks (integer batch counters untouched); comm_stats() reports what GradBuckets moves per step."""
    _init(rank, world, path)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    net.train()
    net(torch.randn(8, 3, 5, 5, generator=torch.Generator().manual_seed(10 + rank)) * (1 + rank))
    mine = net[1].running_var.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    cdist.average_buffers(net)
    ok = torch.allclose(net[1].running_var, (both[0] + both[1]) / 2, rtol=1e-6)
    ok &= not torch.allclose(both[0], both[1])
    ok &= int(net[1].num_batches_tracked) == 1
    red = cdist.GradBuckets(list(net.parameters()), bucket_cap_mb=0.00004)       # ~40 bytes per bucket
    st = red.comm_stats()
    ok &= st['allreduce_bytes_per_step'] == 4 * sum(p.numel() for p in net.parameters())
    ok &= st['buckets'] == len(red.buckets) >= 2 and sum(st['bucket_bytes']) == st['allreduce_bytes_per_step']
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(flag.item() == 1.0))
    dist.destroy_process_group()


@pytest.mark.parametrize('worker', [_worker_global_contrast, _worker_clients_conw, _worker_config2_eight_clients,
                                    _worker_reducer_lifecycle, _worker_adamp_broadcast_state, _worker_rep_gather_buffer,
                                    _worker_average_buffers_and_comm_stats])
def test_two_rank_gloo(worker):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'rdzv')
        procs = [ctx.Process(target=worker, args=(r, 2, path, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0, f'worker exited with {p.exitcode}'
        assert out.get(timeout=10) is True
