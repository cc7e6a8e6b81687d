"""GPU: the multi-rank code paths END TO END on real kernels, with two processes sharing the one GPU of the test box.
RCCL refuses two ranks on one device, so the process group is gloo (it moves device tensors through the host); everything
else -- the HIP kernels, the streams, GradBuckets, the client plan / all-gather, row-sharded con_w, replica re-synchronisation
-- is exactly what runs with one process per GPU over RCCL.

  * test_two_rank_global_contrast_matches_single_process: bench.py's multi-GPU mode in miniature (each rank encodes half of a
    batch, gradient-aware all-gather, full-batch pair loss, bucketed gradient averaging overlapped with backward, fused
    AdamP reading the bucket views) == one process on the whole batch.
  * test_two_rank_round_matches_single_process: one MMFL communication round (2 image + 2 text clients, inter + intra
    contrast, con_w, KD) on 2 ranks == the same round in one process; the two server replicas end identical.
"""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _init(rank, world, path):
    import faulthandler
    import torch.distributed as dist
    # a worker that is still running after 9 minutes prints every thread's stack and exits: a hung collective (or a library stuck
    # in a search) must show WHERE, the parent's join() only sees a missing exit code
    faulthandler.dump_traceback_later(540, exit=True)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    # MIOpen immediate mode in the workers (the product's own switch, creamfl_amd/runtime.py): two processes that share one GPU
    # and both let the library time every solver of every new problem is where a round-5 run of this file sat for nine minutes
    # (one rank inside a convolution's search, the other waiting in the bucket all-reduce); which library kernel runs is not what
    # these tests are about
    os.environ['CFL_MIOPEN_IMMEDIATE'] = '1'
    os.environ.setdefault('MIOPEN_FIND_MODE', '2')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', init_method=f'file://{path}', rank=rank, world_size=world)


def _small_engine(dev, dim=64):
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    torch.manual_seed(0)
    cfg = default_config(embed_dim=dim, cnn_type='resnet18', not_bert=False)
    cfg.model.bert_name = 'bert-mini'
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.model.eval()                    # BatchNorm on running statistics, no dropout: rank-local batches = one big batch
    return eng


def _kd_loss(eng, b, sl):
    """kd_weight * MSE(image features, agg[d_idx]) + the same for captions on rows `sl` of batch b (MMFL.kd_terms)."""
    from creamfl_amd import ops
    g = torch.Generator().manual_seed(99)
    agg_i = torch.nn.functional.normalize(torch.randn(40, 64, generator=g), dim=-1).to(b[0].device)
    agg_t = torch.nn.functional.normalize(torch.randn(40, 64, generator=g), dim=-1).to(b[0].device)
    d_idx = torch.randperm(40, generator=g)[:16].to(b[0].device)[sl]
    out = eng.model(b[0][sl], b[1][sl], None, b[3][sl])
    return ops.kd_mse(out['image_features'], agg_i, d_idx, 0.7) + ops.kd_mse(out['caption_features'], agg_t, d_idx, 0.7)


def _global_contrast_worker(rank, world, path, outdir):
    import torch.distributed as dist
    from creamfl_amd.utils.synthetic import coco_batch
    _init(rank, world, path)
    dev = torch.device('cuda', 0)
    eng = _small_engine(dev)
    eng.enable_data_parallel(bucket_cap_mb=4)            # several buckets
    b = coco_batch(16, dev, seed=7, bert=True, img=64, min_len=12, max_len=12)
    sl = slice(rank * 8, rank * 8 + 8)
    loss, _ = eng.train_step(b[0][sl], b[1][sl], None, b[3][sl])
    torch.cuda.synchronize()
    views = eng.dp.reducer.grad_views()
    named = dict(eng.model.named_parameters())
    keys = ['img_enc.fc.weight', 'img_enc.cnn.conv1.weight', 'img_enc.cnn.layer3.0.conv1.weight', 'linear.weight',
            'txt_enc.encoder.layer.0.attention.self.query.weight' if 'txt_enc.encoder.layer.0.attention.self.query.weight' in named
            else [k for k in named if k.startswith('txt_enc') and k.endswith('weight')][3]]
    rec = {'loss': loss.detach().cpu(), 'n_buckets': len(eng.dp.reducer.buckets),
           'grads': {k: views[named[k]].detach().float().cpu() for k in keys},
           'weights': {k: named[k].detach().float().cpu() for k in keys}}
    # ... followed by a KD step (MMFL.py:346-391) through the same reducer -- ADVICE r2 (high): the round-2 distill loop
    # back-propagated without prepare/finish and the fused optimizer applied the CONTRASTIVE step's stale averages.
    kd_loss = _kd_loss(eng, b, sl)
    crit_steps = [eng.optimizer.state[p]['step'] for p in eng.criterion.parameters()]
    eng.backward_and_step(kd_loss)
    torch.cuda.synchronize()
    rec['kd_weights'] = {k: named[k].detach().float().cpu() for k in keys}
    rec['kd_grads'] = {k: views[named[k]].detach().float().cpu() for k in keys}      # the fused step leaves the averages intact
    # the criterion's scalars got no gradient in the KD step: their step counts lag (per-parameter `step`)
    rec['crit_steps'] = (crit_steps, [eng.optimizer.state[p]['step'] for p in eng.criterion.parameters()],
                         eng.optimizer.state[named[keys[0]]]['step'])
    # and a backward pass that bypasses the reducer must be refused, not silently fed stale averages
    eng.optimizer.zero_grad(set_to_none=True)
    eng.backward(_kd_loss(eng, b, sl))
    try:
        eng.optimizer_step()
        rec['stale_refused'] = False
    except RuntimeError:
        rec['stale_refused'] = True
    if rank == 0:
        torch.save(rec, os.path.join(outdir, 'dp.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, world, *args):
    ctx = mp.get_context('spawn')
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'rdzv')
        procs = [ctx.Process(target=fn, args=(r, world, path) + args) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0, f'worker exited with {p.exitcode}'


def test_two_rank_global_contrast_matches_single_process():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd.utils.synthetic import coco_batch
    with tempfile.TemporaryDirectory() as out:
        _spawn(_global_contrast_worker, 2, out)
        got = torch.load(os.path.join(out, 'dp.pt'))
    dev = torch.device('cuda', 0)
    eng = _small_engine(dev)
    b = coco_batch(16, dev, seed=7, bert=True, img=64, min_len=12, max_len=12)
    # reference gradients of the SAME initial weights on the whole batch (before the optimizer touches them)
    loss, _ = eng.forward_loss(b[0], b[1], None, b[3])
    eng.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    named = dict(eng.model.named_parameters())
    assert got['n_buckets'] > 1
    np.testing.assert_allclose(got['loss'].item(), loss.item(), rtol=1e-4)
    for k, g in got['grads'].items():
        ref = named[k].grad.detach().float().cpu()
        # (library convolutions: which fp32 weight-gradient algorithm MIOpen picks for 8 rows and for 16 depends on what its
        # find-db has recorded by then -- Winograd-class kernels differ by ~1e-3 of scale in the stem's gradient)
        np.testing.assert_allclose(g.numpy(), ref.numpy(), rtol=2e-3, atol=3e-3 * float(ref.abs().max()), err_msg=k)
    # ... and the step itself: same update as the single process
    gref = {k: named[k].grad.detach().float().cpu().clone() for k in got['weights']}
    lr = float(eng.optimizer.param_groups[0]['lr'])
    eng.optimizer_step()
    torch.cuda.synchronize()

    def same_weights(w, ref, atol, what, grad=None, steps=1):
        # The first AdamP step moves every weight by lr * g / (|g| + eps) ~ lr * sign(g): where the library's convolution algorithms
        # (picked per process from whatever its find-db holds by then) round a near-zero gradient to the other side of zero, that
        # ONE element differs by 2 lr.  Element-wise equality for all but a sliver of the tensor, AND -- what makes the sliver
        # an explanation instead of an excuse -- every element that is off (a) differs by no more than
        # 2 lr per step taken and (b), after the first step, sits on a reference gradient small enough for the two runs to
        # disagree about its sign: |g| within the tolerance the gradients themselves were just compared at.
        d = (w - ref).abs()
        off = d > atol + 1e-4 * ref.abs()
        assert float(off.float().mean()) < 2e-3, (what, float(off.float().mean()), float(d.max()))
        if bool(off.any()):
            assert float(d[off].max()) <= 2.2 * lr * steps, (what, 'an off element moved by more than 2 lr per step', float(d[off].max()), lr)
            if grad is not None:
                sign_band = 3e-3 * float(grad.abs().max()) + 2e-3 * grad.abs()
                assert bool((grad.abs()[off] <= sign_band[off]).all()), (
                    what, 'an off element has a gradient too large for a sign flip',
                    float((grad.abs()[off] / float(grad.abs().max())).max()))

    for k, w in got['weights'].items():
        same_weights(w, named[k].detach().float().cpu(), 2e-5, k, grad=gref[k])
    # the KD step after it: mean-MSE over each rank's half, averaged by the reducer == mean-MSE over the whole batch.
    # Compared as UPDATES (weights after - weights before the KD step): a stale-gradient bug replays the contrastive
    # step's direction, which is uncorrelated with the KD direction.
    before = {k: named[k].detach().float().cpu() for k in got['weights']}
    eng.backward_and_step(_kd_loss(eng, b, slice(0, 16)))
    torch.cuda.synchronize()
    for k, g in got['kd_grads'].items():
        ref = named[k].grad.detach().float().cpu()
        # (measured 3e-3 of scale at the stem convolution when the suite's earlier tests had left other algorithms in the find-db;
        # a stale-gradient bug gives an UNCORRELATED direction: the cosine below is the sharp check)
        np.testing.assert_allclose(g.numpy(), ref.numpy(), rtol=5e-3, atol=1e-2 * float(ref.abs().max()), err_msg='KD grad ' + k)
        gc, rc = g.double().flatten(), ref.double().flatten()
        assert float(torch.dot(gc, rc) / (gc.norm() * rc.norm() + 1e-30)) > 0.999, ('KD grad direction', k)
    for k, w in got['kd_weights'].items():
        ref_upd = (named[k].detach().float().cpu() - before[k]).numpy().ravel()
        got_upd = (w - got['weights'][k]).numpy().ravel()
        cos = float(np.dot(ref_upd, got_upd) / (np.linalg.norm(ref_upd) * np.linalg.norm(got_upd) + 1e-30))
        assert cos > 0.99, (k, cos)
        same_weights(w, named[k].detach().float().cpu(), 6e-5, 'after KD ' + k, steps=2)
    before_steps, after_steps, trunk_steps = got['crit_steps']
    assert before_steps == after_steps == [1, 1] and trunk_steps == 2
    assert got['stale_refused'] is True


def _round_args(server_dp=0, rep_wire='fp32'):
    from conftest import reference_main_namespace
    args, _ = reference_main_namespace(name='/tmp/creamfl_test_mr', feature_dim=64, pub_data_num=64, local_epochs=1, comm_rounds=1,
                                       num_img_clients=2, num_txt_clients=2, num_mm_clients=0, client_num_per_round=4,
                                       contrast_local_intra=True, contrast_local_inter=True, cnn_type='resnet18',
                                       bert_name='bert-mini', image_size=64, test_pairs=100, quiet=True, save_checkpoints=False,
                                       server_dp=server_dp, rep_wire=rep_wire)
    return args


def _run_round(seed=20, server_dp=0, rep_wire='fp32'):
    import random
    from creamfl_amd.algorithms.MMFL import MMFL
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    args = _round_args(server_dp, rep_wire)
    algo = MMFL(args, None)
    algo.config.dataloader.batch_size = 32
    algo.config.train.use_fp16 = False
    algo.create_model(args)
    algo.load_dataset(args)
    captured = {}
    orig = algo.aggregation

    def spy(i_vec, t_vec):
        out = orig(i_vec, t_vec)
        captured.update(n_img=len(i_vec), n_txt=len(t_vec), agg_i=out[0].detach().float().cpu(), agg_t=out[1].detach().float().cpu())
        return out

    algo.aggregation = spy
    algo.train(0)
    torch.cuda.synchronize()
    sd = {k: v.detach().float().cpu() for k, v in algo.engine.model.named_parameters()}     # weights (running statistics
    # are activation means: they magnify the weight noise by the activation scale and are not compared)
    trained = {t.client_idx: (getattr(t, "last_contrast_loss", None) is not None) for t in algo.total_local_trainers}
    return captured, sd, trained


def _round_worker(rank, world, path, outdir, server_dp=0, rep_wire='fp32'):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    _init(rank, world, path)
    captured, sd, trained = _run_round(server_dp=server_dp, rep_wire=rep_wire)
    torch.save({'captured': captured, 'sd': sd, 'trained': trained}, os.path.join(outdir, f'round_{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_round_matches_single_process():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as out:
        _spawn(_round_worker, 2, out)
        r0 = torch.load(os.path.join(out, 'round_0.pt'))
        r1 = torch.load(os.path.join(out, 'round_1.pt'))
    # each client was trained on exactly one rank: the one that owns it (client_idx % 2)
    for idx in r0['trained']:
        assert r0['trained'][idx] == (idx % 2 == 0) and r1['trained'][idx] == (idx % 2 == 1), (idx, r0['trained'], r1['trained'])
    # both ranks aggregated all four clients' representations and hold the same aggregate and the same server replica
    assert r0['captured']['n_img'] == 2 and r0['captured']['n_txt'] == 2
    for k in ('agg_i', 'agg_t'):
        np.testing.assert_allclose(r0['captured'][k].numpy(), r1['captured'][k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    # The replicas see the same data and (reseed_from_rank0) the same dropout draws; what is left between them is the
    # run-to-run noise of the library convolution weight gradients (atomics) pushed through AdamP's sign-like normalised
    # update: a few 1e-5 after the ~4 server steps of this round, bounded by a few x lr (2e-4).  MMFL.train re-synchronises
    # the replicas (weights, masters, moments) at the start of every round.
    for k in r0['sd']:
        d = np.abs(r0['sd'][k].numpy() - r1['sd'][k].numpy())
        assert d.max() <= 2e-3, (k, float(d.max()))                              # <= ~10 steps x lr, even where a sign flipped
        assert float((d > 2e-4).mean()) <= 0.1, (k, float((d > 2e-4).mean()))    # and most elements far closer
    # ... and the distributed round is the single-process round
    captured, sd, trained = _run_round()
    assert all(trained.values())
    for k in ('agg_i', 'agg_t'):
        np.testing.assert_allclose(r0['captured'][k].numpy(), captured[k].numpy(), rtol=2e-3, atol=2e-4, err_msg=k)


def test_two_rank_round_data_parallel_server_phases():
    """The same round with the server phases DATA-PARALLEL (MMFL --server_dp 1, the multi-rank default: every rank encodes
    half of each public batch in the global training and in the KD loop, features all-gathered, gradients bucket-averaged)
    and the representations exchanged through the one-collective bf16-wire buffer (--rep_wire bf16): every client still
    trains on exactly one rank, both ranks aggregate the same 2 + 2 representations, and -- the point of averaging gradients
    instead of replicating steps -- the two server replicas end the round IDENTICAL up to the library's atomics noise."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as out:
        _spawn(_round_worker, 2, out, 1, 'bf16')
        r0 = torch.load(os.path.join(out, 'round_0.pt'))
        r1 = torch.load(os.path.join(out, 'round_1.pt'))
    for idx in r0['trained']:
        assert r0['trained'][idx] == (idx % 2 == 0) and r1['trained'][idx] == (idx % 2 == 1)
    assert r0['captured']['n_img'] == 2 and r0['captured']['n_txt'] == 2
    for k in ('agg_i', 'agg_t'):
        assert np.isfinite(r0['captured'][k].numpy()).all()
        np.testing.assert_allclose(r0['captured'][k].numpy(), r1['captured'][k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    for k in r0['sd']:
        d = np.abs(r0['sd'][k].numpy() - r1['sd'][k].numpy())
        assert np.isfinite(r0['sd'][k].numpy()).all()
        assert d.max() <= 2e-3 and float((d > 2e-4).mean()) <= 0.1, (k, float(d.max()))
