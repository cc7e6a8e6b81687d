"""GPU: fused multi-tensor clip + AdamP (csrc/adamp.hip) against the paper restatement (oracle/adamp.py)
driven by torch's clip_grad_norm_ on CPU.  PARITY UNPINNED w.r.t. the adamp package (not installed)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_params(gen):
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 1, 1), (128, 64, 3, 3), (300, 77), (5, 1030), (10,), (1,), (1,),
              (2, 40000), (33, 7)]
    return [torch.randn(*s, generator=gen) * 0.1 for s in shapes]


@pytest.mark.parametrize('weight_decay,nesterov,channels_last', [(0.0, False, False), (0.01, False, True), (0.01, True, False)])
def test_fused_adamp_matches_oracle(weight_decay, nesterov, channels_last):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from oracle.adamp import AdamP as OracleAdamP
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(0)
    init = _make_params(gen)
    cpu = [torch.nn.Parameter(t.clone()) for t in init]
    gpu = []
    for t in init:
        q = t.clone().to(dev)
        if channels_last and q.dim() == 4:
            q = q.contiguous(memory_format=torch.channels_last)
        gpu.append(torch.nn.Parameter(q))
    n_clip = len(init) - 2                      # the last two tensors play the criterion's (unclipped) scalars
    ocpu = OracleAdamP(cpu, lr=1e-2, weight_decay=weight_decay, nesterov=nesterov)
    ogpu = AdamP(gpu, lr=1e-2, weight_decay=weight_decay, nesterov=nesterov)
    for it in range(4):
        grads = [torch.randn(t.shape, generator=gen) * (3.0 if it % 2 == 0 else 0.01) for t in init]
        # make some tensors scale-invariant-looking: gradient orthogonal to the weight (projection fires)
        for k in (0, 3, 4):
            w = cpu[k].detach().reshape(cpu[k].shape[0], -1)
            g = grads[k].reshape(w.shape)
            g = g - w * (g * w).sum(1, keepdim=True) / (w * w).sum(1, keepdim=True)
            grads[k] = g.reshape(grads[k].shape)
        w = cpu[9].detach().reshape(1, -1)
        g = grads[9].reshape(1, -1)
        grads[9] = (g - w * (g * w).sum() / (w * w).sum()).reshape(grads[9].shape)       # layer-wise only
        for p, q, g in zip(cpu, gpu, grads):
            p.grad = g.clone()
            gg = g.to(dev)
            if channels_last and gg.dim() == 4:
                gg = gg.contiguous(memory_format=torch.channels_last)
            q.grad = gg
        norm = torch.nn.utils.clip_grad_norm_(cpu[:n_clip], 2.0)
        ocpu.step()
        ogpu.step(clip=(gpu[:n_clip], 2.0))
        np.testing.assert_allclose(ogpu.last_grad_norm.item(), norm.item(), rtol=1e-5)
        for k, (p, q) in enumerate(zip(cpu, gpu)):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-4, atol=2e-6,
                                       err_msg=f'step {it} tensor {k} {tuple(p.shape)}')
    # state_dict layout like adamp.AdamP
    st = ogpu.state[gpu[0]]
    assert set(st) == {'step', 'exp_avg', 'exp_avg_sq'} and st['step'] == 4


def test_fused_adamp_bf16_weights_fp32_masters():
    """apex-O2 style: bf16 model weights + bf16 grads, fp32 masters and moments inside the optimizer.  The master
    trajectory must equal the fp32 oracle fed with the same (bf16-valued) gradients; the bf16 shadow must be the
    rounded master."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from oracle.adamp import AdamP as OracleAdamP
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(3)
    init = _make_params(gen)
    cpu = [torch.nn.Parameter(t.clone()) for t in init]
    gpu = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ogpu = AdamP(gpu, lr=1e-2, weight_decay=0.01)
    half = [0, 2, 3, 4, 6]                       # these become bf16 weights
    for k in half:
        ogpu.make_master(gpu[k])
        gpu[k].data = gpu[k].data.to(torch.bfloat16)
    ocpu = OracleAdamP(cpu, lr=1e-2, weight_decay=0.01)
    for it in range(3):
        grads = [torch.randn(t.shape, generator=gen) for t in init]
        for k in half:
            grads[k] = grads[k].to(torch.bfloat16).float()          # the values a bf16 backward would produce
        for k, (p, q, g) in enumerate(zip(cpu, gpu, grads)):
            p.grad = g.clone()
            q.grad = g.to(dev).to(q.dtype)
        norm = torch.nn.utils.clip_grad_norm_(cpu, 2.0)
        ocpu.step()
        ogpu.step(clip=(gpu, 2.0))
        np.testing.assert_allclose(ogpu.last_grad_norm.item(), norm.item(), rtol=1e-5)
        for k, (p, q) in enumerate(zip(cpu, gpu)):
            master = ogpu.state[q]['master'] if k in half else q
            np.testing.assert_allclose(master.detach().float().cpu().numpy(), p.detach().numpy(), rtol=2e-4, atol=2e-6,
                                       err_msg=f'step {it} tensor {k}')
            if k in half:
                assert q.dtype == torch.bfloat16
                assert torch.equal(q.detach(), master.to(torch.bfloat16))


def test_fused_adamp_async_uploads_do_not_race():
    """The gradient-pointer table is uploaded asynchronously from pinned memory while the host runs ahead of the
    GPU; a step must still see ITS pointers (regression: a single staging buffer was overwritten by the next
    step's pointers before the previous copy had executed)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    dev = torch.device('cuda:0')

    def run(sync_every_step):
        gen = torch.Generator().manual_seed(3)
        ps = [torch.nn.Parameter((torch.randn(*s, generator=gen) * 0.1).to(dev)) for s in [(96, 200), (50,), (8, 4, 3, 3)]]
        opt = AdamP(ps, lr=1e-2)
        grads = [[(torch.randn(p.shape, generator=gen)).to(dev) for p in ps] for _ in range(12)]   # distinct addresses
        big = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        for gs in grads:
            for _ in range(6):
                big = (big @ big).clamp_(-1, 1)       # keep the GPU busy so that the host runs ahead
            for p, g in zip(ps, gs):
                p.grad = g
            opt.step(clip=(ps, 2.0))
            if sync_every_step:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [p.detach().cpu().numpy() for p in ps]

    ref = run(True)
    got = run(False)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)


def test_fused_adamp_per_parameter_steps_lagging_gradients():
    """parameters whose gradient is None in some steps keep their own step count (adamp==0.3.0 keeps
    `state['step']` per parameter) -- the `tsteps` branch of optimizers.AdamP.step / CflTensorMeta.step in csrc/adamp.hip.
    Schedule: the two trailing scalars (the criterion's shift / negative_scale) and one matrix get a gradient only on some
    steps, exactly what every KD phase does to the criterion's scalars; the oracle uses per-parameter counts too."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from oracle.adamp import AdamP as OracleAdamP
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(11)
    init = _make_params(gen)
    cpu = [torch.nn.Parameter(t.clone()) for t in init]
    gpu = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    n = len(init)
    # two groups, like the engine's single group + a second one, so that group-local uniformity is exercised as well
    ocpu = OracleAdamP([{'params': cpu[:4]}, {'params': cpu[4:]}], lr=1e-2, weight_decay=0.01)
    ogpu = AdamP([{'params': gpu[:4]}, {'params': gpu[4:]}], lr=1e-2, weight_decay=0.01)
    lag = {n - 1: [1, 0, 0, 1, 0, 1], n - 2: [1, 0, 0, 1, 0, 1], 4: [1, 1, 0, 1, 0, 0], 1: [0, 1, 1, 1, 0, 1]}
    for it in range(6):
        grads = [torch.randn(t.shape, generator=gen) for t in init]
        for k, (p, q, g) in enumerate(zip(cpu, gpu, grads)):
            on = lag.get(k, [1] * 6)[it]
            p.grad = g.clone() if on else None
            q.grad = g.to(dev) if on else None
        clip_cpu = [p for p in cpu[:n - 2]]
        norm = torch.nn.utils.clip_grad_norm_([p for p in clip_cpu if p.grad is not None], 2.0)
        ocpu.step()
        ogpu.step(clip=(gpu[:n - 2], 2.0))
        np.testing.assert_allclose(ogpu.last_grad_norm.item(), norm.item(), rtol=1e-5)
        for k, (p, q) in enumerate(zip(cpu, gpu)):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-4, atol=2e-6,
                                       err_msg=f'step {it} tensor {k} {tuple(p.shape)}')
    for k, (p, q) in enumerate(zip(cpu, gpu)):
        want = sum(lag.get(k, [1] * 6))
        assert ogpu.state[q]['step'] == want == ocpu.state[p]['step'], (k, ogpu.state[q]['step'], want)


def test_fused_adamp_checkpoint_round_trip_is_bit_exact():
    """save -> load -> one step == continuing without the reload, bit for bit, with bf16 trunk weights + fp32
    masters (AdamP.load_state_dict / master_state_dict / load_model_weights / refresh_masters)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import copy
    import io
    from creamfl_amd.algorithms.optimizers import AdamP
    dev = torch.device('cuda:0')

    def build():
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.Flatten(), torch.nn.Linear(16 * 36, 24),
                                  torch.nn.LayerNorm(24)).to(dev).to(memory_format=torch.channels_last)
        opt = AdamP(list(net.parameters()), lr=1e-2, weight_decay=0.01)
        for mod in (net[0], net[2]):
            opt.make_master(mod.weight)
            mod.weight.data = mod.weight.data.to(torch.bfloat16)
        return net, opt

    def grads_for(net, it):
        g = torch.Generator().manual_seed(100 + it)
        for p in net.parameters():
            p.grad = torch.randn(p.shape, generator=g).to(dev).to(p.dtype).contiguous(
                memory_format=torch.channels_last if p.dim() == 4 else torch.contiguous_format)

    net_a, opt_a = build()
    for it in range(2):
        grads_for(net_a, it)
        opt_a.step(clip=(list(net_a.parameters()), 2.0))
    buf = io.BytesIO()
    torch.save({'model': opt_a.master_state_dict(net_a), 'optimizer': opt_a.state_dict()}, buf)
    # A continues
    grads_for(net_a, 2)
    opt_a.step(clip=(list(net_a.parameters()), 2.0))
    # B is rebuilt from the checkpoint (through the CPU, like retrieval_trainer.load_models) and takes the same step
    buf.seek(0)
    ck = torch.load(buf, map_location='cpu')
    assert ck['model']['0.weight'].dtype == torch.float32            # checkpoints hold the masters
    net_b, opt_b = build()
    opt_b.load_state_dict(ck['optimizer'])
    net_b.load_state_dict(ck['model'])                               # casts the fp32 master into the bf16 weight
    named = dict(net_b.named_parameters())
    opt_b.refresh_masters({named[k]: v.to(dev) for k, v in ck['model'].items() if named[k].dtype == torch.bfloat16})
    for p in (net_b[0].weight, net_b[2].weight):
        st = opt_b.state[p]
        assert st['master'].dtype == st['exp_avg'].dtype == torch.float32 and st['exp_avg'].stride() == p.stride()
    grads_for(net_b, 2)
    opt_b.step(clip=(list(net_b.parameters()), 2.0))
    torch.cuda.synchronize()
    for (k, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        assert torch.equal(pa.detach(), pb.detach()), k
        for key in ('master', 'exp_avg', 'exp_avg_sq'):
            if key in opt_a.state[pa]:
                assert torch.equal(opt_a.state[pa][key], opt_b.state[pb][key]), (k, key)
        assert opt_a.state[pa]['step'] == opt_b.state[pb]['step'] == 3


def _graph_toy(dev, seed=0):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(24, 48), torch.nn.Tanh(), torch.nn.Linear(48, 8)).to(dev)
    scal = torch.nn.Parameter(torch.tensor([0.7], device=dev))          # plays the criterion's scalars: skipped by some steps
    return net, scal


def test_fused_adamp_step_inside_a_hip_graph_equals_eager():
    """The fused AdamP inside a replayed HIP graph (graphs.GraphedStep(optimizer=...), the multi-modal client's contrast step,
    MMClientTrainer.py:150-224): the step count of the bias corrections is read from the device (cfl_adamp_step_counted: counter +
    per-tensor offset), the tensor table with the captured gradient addresses is re-uploaded by a node of the graph.  Against the
    same 12 steps run eagerly: 3 eager warm-ups, the capture, replays, ONE eager step on a ragged batch in between (it uploads its
    own table and moves the counter), more replays.  Before them 2 steps that skip the scalar parameter, so the tensors' step
    counts differ (per-tensor offsets).  Weights equal to 1e-6, and state_dict() reports the same step counts."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from creamfl_amd.graphs import GraphedStep
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(5)
    xs = [torch.randn(16, 24, generator=gen).to(dev) for _ in range(12)]
    ragged = torch.randn(5, 24, generator=gen).to(dev)

    def run(graph):
        net, scal = _graph_toy(dev)
        opt = AdamP(list(net.parameters()) + [scal], lr=1e-2, weight_decay=0.01)

        def step(x, with_scalar=True):
            opt.zero_grad(set_to_none=True)
            y = net(x)
            loss = (y * y).mean() * (scal.sum() if with_scalar else 1.0)
            loss.backward()
            opt.step(clip=(net.parameters(), 0.5))
            return loss.detach()
        for x in xs[:2]:
            step(x, with_scalar=False)                                  # the scalar lags two steps behind from here on
        gs = GraphedStep(step, warmup=3, enabled=graph, optimizer=opt)
        losses = []
        for i, x in enumerate(xs[2:]):
            losses.append(float(gs(x)))
            if i == 6:
                losses.append(float(gs(ragged)))                        # another shape: eager, between two replays
        torch.cuda.synchronize()
        sd = opt.state_dict()
        steps = [sd['state'][k]['step'] for k in sorted(sd['state'])]
        return gs, [p.detach().cpu() for p in list(net.parameters()) + [scal]], steps, losses

    gs_e, w_e, steps_e, loss_e = run(False)
    gs_g, w_g, steps_g, loss_g = run(True)
    assert gs_g.failed is None, gs_g.failed
    assert gs_g.replays == 7 and gs_g.calls == 11 and gs_e.replays == 0   # 3 warm-ups, capture + 6 replays, one ragged call
    assert steps_e == steps_g == [13, 13, 13, 13, 11]
    np.testing.assert_allclose(loss_g, loss_e, rtol=1e-5)
    for a, b in zip(w_e, w_g):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-5, atol=1e-6)


def test_fused_adamp_captured_step_goes_stale_when_its_parameters_are_skipped():
    """A captured step's table holds step-count OFFSETS from the device counter.  An eager step that skips one of its parameters
    (the KD step skips the criterion's scalars, MMFL.py:346-391) moves the counter without that parameter: the handle reports
    stale, GraphedStep drops the graph and the calls go on eagerly -- with the same weights as an all-eager run."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from creamfl_amd.graphs import GraphedStep
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(6)
    xs = [torch.randn(16, 24, generator=gen).to(dev) for _ in range(9)]

    def run(graph):
        net, scal = _graph_toy(dev, 1)
        opt = AdamP(list(net.parameters()) + [scal], lr=1e-2)

        def step(x, with_scalar=True):
            opt.zero_grad(set_to_none=True)
            loss = (net(x) ** 2).mean() * (scal.sum() if with_scalar else 1.0)
            loss.backward()
            opt.step()
            return loss.detach()
        msgs = []
        gs = GraphedStep(step, warmup=2, enabled=graph, optimizer=opt, log=msgs.append)
        for x in xs[:5]:
            gs(x)                                                       # 2 warm-ups, capture, 2 replays
        step(xs[5], with_scalar=False)                                  # skips the scalar
        for x in xs[6:]:
            gs(x)
        torch.cuda.synchronize()
        return gs, msgs, [p.detach().cpu() for p in list(net.parameters()) + [scal]], opt.state_dict()

    gs_e, _, w_e, sd_e = run(False)
    gs_g, msgs, w_g, sd_g = run(True)
    assert gs_g.replays == 3 and gs_g.graph is None and len(msgs) == 1 and 'parameter set' in msgs[0]
    assert [sd_g['state'][k]['step'] for k in sorted(sd_g['state'])] == [9, 9, 9, 9, 8]
    for a, b in zip(w_e, w_g):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-5, atol=1e-6)


def test_fused_adamp_failed_capture_rolls_the_step_counts_back():
    """a capture that fails AFTER optimizer.step() was recorded (here: the captured function raises behind it) must not
    leave the host's counts one ahead of the device counter -- nothing recorded ever ran.  The step is re-run eagerly by GraphedStep,
    counted once; the run ends with the weights and step counts of an all-eager run, and a LATER graph on the same optimizer (next
    round's client graph, the server's KD graph) computes its offsets from a host count that equals the device's."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.algorithms.optimizers import AdamP
    from creamfl_amd.graphs import GraphedStep
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(8)
    xs = [torch.randn(16, 24, generator=gen).to(dev) for _ in range(10)]

    def run(mode):
        net, scal = _graph_toy(dev, 2)
        opt = AdamP(list(net.parameters()) + [scal], lr=1e-2, weight_decay=0.01)
        state = {'boom': mode == 'fail'}

        def step(x):
            opt.zero_grad(set_to_none=True)
            loss = (net(x) ** 2).mean() * scal.sum()
            loss.backward()
            opt.step(clip=(net.parameters(), 0.5))
            if state['boom'] and torch.cuda.is_current_stream_capturing():
                state['boom'] = False
                raise RuntimeError('injected: the capture fails behind optimizer.step()')
            return loss.detach()
        msgs = []
        gs = GraphedStep(step, warmup=2, enabled=mode != 'eager', optimizer=opt, log=msgs.append)
        for x in xs[:5]:
            gs(x)
        if mode == 'fail':
            assert gs.failed is not None and 'injected' in gs.failed and len(msgs) == 1 and gs.replays == 0
            assert opt._gstep_host == int(opt._gstep_dev) == 5
        gs2 = GraphedStep(step, warmup=1, enabled=mode != 'eager', optimizer=opt)       # a later graph on the same optimizer
        for x in xs[5:]:
            gs2(x)
        torch.cuda.synchronize()
        if mode == 'fail':
            assert gs2.failed is None and gs2.replays == 4
            assert opt._gstep_host == int(opt._gstep_dev) == 10
        sd = opt.state_dict()
        return [p.detach().cpu() for p in list(net.parameters()) + [scal]], [sd['state'][k]['step'] for k in sorted(sd['state'])]

    w_e, steps_e = run('eager')
    w_f, steps_f = run('fail')
    assert steps_e == steps_f == [10] * 5
    for a, b in zip(w_e, w_f):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-5, atol=1e-6)


def test_graphed_step_refuses_a_capture_while_a_loss_is_alive():
    """a loss of an earlier eager step that is still referenced keeps the AccumulateGrad nodes of the step's parameters
    alive; capturing then can FAULT inside hipStreamEndCapture (graphs.py).  The precondition is checked: the step stays eager and
    says why, instead of relying on the callers' discipline."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.graphs import GraphedStep, live_grad_nodes
    dev = torch.device('cuda:0')
    net, _ = _graph_toy(dev, 3)
    opt = torch.optim.SGD(net.parameters(), lr=1e-2)
    x = torch.randn(16, 24, device=dev)

    def step(x):
        opt.zero_grad(set_to_none=True)
        loss = (net(x) ** 2).mean()
        loss.backward()
        opt.step()
        return loss.detach()
    kept = (net(x) ** 2).mean()                       # what a logging line / a closure / a debugger would hold on to
    assert live_grad_nodes(net.parameters()) == 4
    msgs = []
    gs = GraphedStep(step, warmup=1, optimizer=opt, log=msgs.append)
    for _ in range(3):
        gs(x)
    assert gs.graph is None and gs.replays == 0 and len(msgs) == 1 and 'AccumulateGrad' in msgs[0]
    del kept
    assert live_grad_nodes(net.parameters()) == 0
    gs = GraphedStep(step, warmup=1, optimizer=opt)
    for _ in range(3):
        gs(x)
    torch.cuda.synchronize()
    assert gs.failed is None and gs.replays == 2
