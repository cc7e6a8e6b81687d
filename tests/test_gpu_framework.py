"""GPU: the host mirror of the reference API end to end -- server step vs. the CPU oracle port, evaluator vs.
the recall oracle, and one full communication round (BASELINE.json configs[0]-style plumbing) on tiny
synthetic data."""
import copy
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle
import oracle.step as ostep

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _small_cfg(dim=64, cnn='resnet18', not_bert=False):
    from creamfl_amd.utils.config import default_config
    cfg = default_config(embed_dim=dim, cnn_type=cnn, not_bert=not_bert)
    cfg.model.bert_name = 'bert-mini'
    return cfg


def test_server_step_matches_cpu_port(dev):
    """TrainerEngine.train_step (fp32 trunks, HIP head + loss + clip + AdamP) against oracle.step on identical
    weights and batch: features, loss and the updated parameters."""
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.synthetic import coco_batch
    from oracle.adamp import AdamP as OracleAdamP
    torch.manual_seed(0)
    cfg = _small_cfg()
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model.eval()                                   # dropout off / BN in eval mode: deterministic on both sides
    cpu_model = copy.deepcopy(eng.model)
    crit = SimpleNamespace(negative_scale=torch.nn.Parameter(torch.tensor([15.0])),
                           shift=torch.nn.Parameter(torch.tensor([15.0])))
    eng.model_to_device()
    b = coco_batch(24, 'cpu', seed=5, bert=True, img=96)
    # --- CPU oracle port
    params = [p for p in cpu_model.parameters()] + [crit.negative_scale, crit.shift]
    opt = OracleAdamP(params, lr=cfg.optimizer.learning_rate)
    img_c, txt_c = ostep.pcme_forward_cpu(cpu_model, b[0], b[1], b[3])
    loss_c, _ = ostep.contrastive_step_cpu(cpu_model, crit, opt, b, cfg.train.grad_clip)
    # --- HIP path
    out = eng.model(b[0].to(dev), b[1].to(dev), None, b[3].to(dev))
    assert list(out.keys()) == ['image_features', 'image_attentions', 'image_residuals', 'image_logsigma',
                                'image_logsigma_att', 'caption_features', 'caption_attentions', 'caption_residuals',
                                'caption_logsigma', 'caption_logsigma_att']
    np.testing.assert_allclose(out['image_features'].detach().cpu().numpy(), img_c.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out['caption_features'].detach().cpu().numpy(), txt_c.detach().numpy(), rtol=2e-4, atol=2e-5)
    loss_g, ld = eng.train_step(b[0].to(dev), b[1].to(dev), None, b[3].to(dev))
    np.testing.assert_allclose(loss_g.item(), loss_c.item(), rtol=1e-4)        # north_star: loss within 1e-4 (measured ~2e-7)
    assert set(ld.keys()) >= {'i2t_loss', 't2i_loss', 'loss', 'shift', 'negative_scale'}
    np.testing.assert_allclose(ld['loss'], loss_g.item(), rtol=1e-6)
    # gradients (what the optimizer consumed): the CPU side holds them clipped in place (clip_grad_norm_), the HIP side applies
    # the same coefficient inside the fused optimizer and leaves p.grad raw -- compare after scaling
    gp_ = dict(eng.model.named_parameters())
    cp_ = dict(cpu_model.named_parameters())
    tot = torch.sqrt(sum((p.grad.detach().double() ** 2).sum() for p in eng.model.parameters() if p.grad is not None)).item()
    coef = min(1.0, cfg.train.grad_clip / (tot + 1e-6))
    for n in ['img_enc.fc.weight', 'img_enc.pie_net.attention.w_1.weight', 'img_enc.pie_net.attention.w_2.weight', 'linear.weight',
              'img_enc.cnn.conv1.weight', 'img_enc.cnn.layer2.0.conv1.weight', 'img_enc.pie_net.layer_norm.weight']:
        g_hip = gp_[n].grad.detach().float().cpu().numpy() * coef
        g_cpu = cp_[n].grad.detach().numpy()
        np.testing.assert_allclose(g_hip, g_cpu, rtol=5e-3, atol=5e-4 * float(np.abs(g_cpu).max()), err_msg=n)
    # updated parameters: one AdamP step moves every weight by ~lr (at step 1 the update is lr * sign-like: elements whose gradient
    # is ~0 may flip, hence a direction check here; the optimizer itself is checked element-wise in test_gpu_optimizer.py)
    names = ['img_enc.fc.weight', 'img_enc.pie_net.attention.w_1.weight', 'linear.weight', 'img_enc.cnn.conv1.weight']
    gp = dict(eng.model.named_parameters())
    cp = dict(cpu_model.named_parameters())
    torch.manual_seed(0)
    ref0 = dict(TrainerEngineInit(cfg).named_parameters())
    for n in names:
        dg = gp[n].detach().cpu() - ref0[n]
        dc = cp[n].detach() - ref0[n]
        cos = torch.nn.functional.cosine_similarity(dg.flatten(), dc.flatten(), dim=0).item()
        assert cos > 0.98, (n, cos)
    np.testing.assert_allclose(eng.criterion.shift.item(), crit.shift.item(), rtol=1e-4)
    np.testing.assert_allclose(eng.criterion.negative_scale.item(), crit.negative_scale.item(), rtol=1e-4)


def test_server_step_train_mode_stage_by_stage(dev):
    """VERDICT r2 weak #2.  The SAME step in train() mode (BatchNorm on batch statistics, running statistics updated; dropout
    probability 0 so that both sides are deterministic), fp32 trunks, attributed stage by stage against the CPU port:
      trunk   -- the ResNet map and the BERT [CLS] state (library convolutions / GEMMs on both sides: accumulation order only)
      head    -- PIE head + l2norm given the CPU port's OWN trunk output (isolates csrc/pie*.hip)
      loss    -- pair loss given the CPU port's OWN features (isolates csrc/pair_loss.hip)
      step    -- the loss of the whole step, the clipped gradients, and the parameter UPDATE element by element
    Each stage has its own bound; the whole-step loss is held to 2e-4 (north_star: 1e-4 on the loss given identical inputs --
    that is the `loss` stage, held to 2e-5 -- the rest is fp32 accumulation-order noise of the library trunks, measured and
    printed)."""
    from creamfl_amd import ops
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.synthetic import coco_batch
    from oracle.adamp import AdamP as OracleAdamP
    torch.manual_seed(0)
    cfg = _small_cfg()
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    for m in eng.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    eng.model.train()
    cpu_model = copy.deepcopy(eng.model).train()
    crit = SimpleNamespace(negative_scale=torch.nn.Parameter(torch.tensor([15.0])),
                           shift=torch.nn.Parameter(torch.tensor([15.0])))
    eng.model_to_device()
    b = coco_batch(24, 'cpu', seed=5, bert=True, img=96)
    bg = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    report = {}

    def rel(got, want):
        got, want = got.detach().double().cpu(), want.detach().double().cpu()
        return float((got - want).abs().max() / want.abs().max())

    # ---- stage: trunks (no parameter update yet; BatchNorm running statistics are restored afterwards on both sides)
    state_g = copy.deepcopy(eng.model.state_dict())
    state_c = copy.deepcopy(cpu_model.state_dict())
    with torch.no_grad():
        fmap_c = cpu_model.img_enc.cnn.features(b[0])
        fmap_g = eng.model.img_enc.cnn.features(bg[0])
        cls_c = cpu_model.txt_enc(**cpu_model._bert_inputs(b[1], None, b[3]))['last_hidden_state'][:, 0]
        cls_g = eng.model.txt_enc(**eng.model._bert_inputs(bg[1], None, bg[3]))['last_hidden_state'][:, 0]
    report['trunk_image_map'] = rel(fmap_g.float(), fmap_c)
    report['trunk_text_cls'] = rel(cls_g.float(), cls_c)
    eng.model.load_state_dict(state_g)
    cpu_model.load_state_dict(state_c)
    # ---- stage: head given the CPU trunk output
    img_c, txt_c = ostep.pcme_forward_cpu(cpu_model, b[0], b[1], b[3])
    cpu_model.load_state_dict(state_c)
    with torch.no_grad():
        enc = eng.model.img_enc
        fm = fmap_c.to(dev)
        head_g = enc.head(fm)[0]
    report['head_given_cpu_trunk'] = rel(head_g, img_c)
    # ---- stage: loss given the CPU features
    lg, _ = ops.pair_loss(img_c.detach().to(dev), txt_c.detach().to(dev), torch.tensor([15.0], device=dev),
                          torch.tensor([15.0], device=dev))
    lc, _ = oracle.pair_loss_literal(img_c.detach(), txt_c.detach(), torch.tensor([15.0]), torch.tensor([15.0]))
    cf = oracle.pair_loss_closed_form(img_c.detach(), txt_c.detach(), 15.0, 15.0)
    report['loss_given_cpu_features_vs_fp64'] = abs(lg.item() - float(cf['loss'])) / abs(float(cf['loss']))
    report['reference_fp32_loss_vs_fp64'] = abs(lc.item() - float(cf['loss'])) / abs(float(cf['loss']))
    # ---- the whole step on both sides
    params = [p for p in cpu_model.parameters()] + [crit.negative_scale, crit.shift]
    opt = OracleAdamP(params, lr=cfg.optimizer.learning_rate)
    w0 = {n: p.detach().clone() for n, p in cpu_model.named_parameters()}
    loss_c, _ = ostep.contrastive_step_cpu(cpu_model, crit, opt, b, cfg.train.grad_clip)
    loss_g, _ = eng.train_step(bg[0], bg[1], None, bg[3])
    report['step_loss'] = abs(loss_g.item() - loss_c.item()) / abs(loss_c.item())
    gp_ = dict(eng.model.named_parameters())
    cp_ = dict(cpu_model.named_parameters())
    tot = torch.sqrt(sum((p.grad.detach().double() ** 2).sum() for p in eng.model.parameters() if p.grad is not None)).item()
    coef = min(1.0, cfg.train.grad_clip / (tot + 1e-6))
    names = ['img_enc.fc.weight', 'img_enc.pie_net.attention.w_1.weight', 'linear.weight', 'img_enc.cnn.conv1.weight',
             'img_enc.cnn.layer2.0.conv1.weight', 'img_enc.cnn.layer4.1.bn2.weight', 'img_enc.pie_net.layer_norm.weight']
    lr = cfg.optimizer.learning_rate
    for n in names:
        g_hip = gp_[n].grad.detach().float().cpu() * coef
        g_cpu = cp_[n].grad.detach()
        report['grad ' + n] = rel(g_hip, g_cpu)
        # the update, element by element, where the gradient is not noise: at step 1 AdamP moves an element by lr * g / (|g| +
        # eps'), so elements whose gradient is ~0 are ill-conditioned on BOTH sides and are left out (they are < 15 %)
        upd_g = gp_[n].detach().float().cpu() - w0[n]
        upd_c = cp_[n].detach() - w0[n]
        sig = g_cpu.abs() > 1e-2 * g_cpu.abs().max()
        assert float(sig.float().mean()) > 0.5, (n, float(sig.float().mean()))
        report['update ' + n] = float((upd_g - upd_c)[sig].abs().max() / lr)
    # BatchNorm running statistics moved identically
    report['bn_running_mean'] = rel(eng.model.img_enc.cnn.bn1.running_mean, cpu_model.img_enc.cnn.bn1.running_mean)
    print('\nstage-by-stage residuals (max |diff| / max |ref|; updates in units of lr):')
    for k, v in report.items():
        print('  %-55s %.3e' % (k, v))
    try:                                                       # kept with the round's evidence when run through gpurun
        import json
        import os
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(report, open('gpurun_out/s1_stage_residuals.json', 'w'), indent=1)
    except OSError:
        pass
    assert report['trunk_image_map'] < 5e-5 and report['trunk_text_cls'] < 1e-5        # measured 3.9e-6 / 3.7e-7
    assert report['head_given_cpu_trunk'] < 1e-5                                        # measured 6.8e-7
    assert report['loss_given_cpu_features_vs_fp64'] < 2e-6                             # measured 5.9e-8 (reference fp32: 1.6e-7)
    assert report['step_loss'] < 2e-5                                                   # measured 2.1e-7 (north_star: 1e-4)
    assert report['bn_running_mean'] < 1e-4
    for k, v in report.items():
        # library convolution weight gradients (split-K / atomics, the deepest accumulation of the step) carry the residual:
        # 1e-3 .. 7e-3 at conv1 from lease to lease; everything computed by this repo's kernels or by GEMMs is at 1e-5
        deep = '.cnn.conv' in k or '.cnn.layer' in k and 'conv' in k
        if k.startswith('grad '):
            assert v < (3e-2 if deep else 2e-4), (k, v)
        if k.startswith('update '):                # a wrong / stale gradient moves an element by ~lr or 2 lr
            assert v < (0.5 if deep else 1e-3), (k, v)


def TrainerEngineInit(cfg):
    """the initial weights again (same seed, same construction order as TrainerEngine.create)"""
    from creamfl_amd.networks.models import get_model
    return get_model({'<pad>': 0}, cfg.model, False)


def test_evaluator_scores_match_oracle(dev):
    from creamfl_amd.algorithms.eval_coco import COCOEvaluator
    from creamfl_amd.networks.models import get_model
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    torch.manual_seed(1)
    cfg = _small_cfg(dim=32)
    model = get_model({'<pad>': 0}, cfg.model, False).to(dev).eval()
    ev = COCOEvaluator(eval_method='matmul', verbose=False, eval_device=str(dev), extract_device=str(dev), n_crossfolds=5)
    ev.set_model(model)
    loader = SyntheticCocoLoader(250, 50, seed=3, bert=True, captions_per_image=5, device='cpu', img=64)
    scores = ev.evaluate(loader, n_crossfolds=5, n_images_per_crossfold=10, n_captions_per_crossfold=50)
    assert set(scores) == {'mean_log_image_sigma', 'mean_log_caption_sigma', 'n_fold', 'i2t', 't2i', 'rsum', 'medr', 'meanr'}
    assert set(scores['i2t']) == {'recall_1', 'recall_5', 'recall_10', 'rsum', 'medr', 'meanr'}
    ef = ev.extract_features(loader)
    img, cap = ef['image_features'].cpu().numpy(), ef['caption_features'].cpu().numpy()
    icls, ccls = ef['image_classes'].numpy(), ef['caption_classes'].numpy()
    assert img.shape == (50, 32) and cap.shape == (250, 32)
    # a randomly initialised encoder maps all inputs to nearly the same point, so similarities can tie to within
    # fp64 round-off; ranks are then only defined up to the tie (the reference's unstable sort has the same
    # ambiguity).  Check the kernel's ranks against oracle bounds with a 1e-12 tie window, and the evaluator's
    # aggregation (recall / medr / meanr) exactly on those ranks.
    from creamfl_amd import ops
    for (q, g, ql, gl, key) in [(img, cap, icls, ccls, 'i2t'), (cap, img, ccls, icls, 't2i')]:
        ranks = ops.rank_count(torch.from_numpy(q).to(dev), torch.from_numpy(g).to(dev), ql, gl).cpu().numpy()
        sims = q.astype(np.float64) @ g.astype(np.float64).T
        pos = ql[:, None] == gl[None, :]
        best = np.where(pos, sims, -np.inf).max(1)
        lo = (sims > best[:, None] + 1e-12).sum(1)
        hi = ((sims > best[:, None] - 1e-12) & ~pos).sum(1)
        assert np.all(ranks >= lo) and np.all(ranks <= hi)
        want = oracle.recall_scores(ranks.astype(np.float64))
        for k, v in want.items():
            assert scores[key][k] == v, (key, k)


def test_one_communication_round(dev):
    """config[0]-style plumbing: 1 image + 1 text + 1 multimodal client, inter + intra contrast, con_w, KD, eval."""
    from creamfl_amd.algorithms.MMFL import MMFL
    torch.manual_seed(2)
    M = 96
    args = SimpleNamespace(name='/tmp/creamfl_test', feature_dim=64, pub_data_num=M, not_bert=False, mlp_local=False,
                           server_lr=2e-4, local_epochs=1, comm_rounds=1, num_img_clients=1, num_txt_clients=1,
                           num_mm_clients=1, client_num_per_round=3, agg_method='con_w', contrast_local_intra=True,
                           contrast_local_inter=True, interintra_weight=0.5, loss_scale=False, kd_weight=0.3,
                           disable_distill=False, save_client=False, device=0, cnn_type='resnet18', bert_name='bert-mini',
                           image_size=64, test_pairs=100, quiet=True, save_checkpoints=False)
    algo = MMFL(args, None)
    algo.config.dataloader.batch_size = 32
    algo.config.train.use_fp16 = False
    algo.create_model(args)
    algo.load_dataset(args)
    captured = {}
    orig = algo.aggregation

    def spy(i_vec, t_vec):
        captured['i'] = [v.clone() for v in i_vec]
        captured['t'] = [v.clone() for v in t_vec]
        captured['g_txt'] = algo.global_txt_feature.clone()
        out = orig(i_vec, t_vec)
        captured['agg_i'] = out[0].clone()
        return out

    algo.aggregation = spy
    algo.train(0)
    assert algo.global_img_feature.shape == (M, 64) and algo.global_img_feature.is_cuda
    assert len(captured['i']) == 2 and len(captured['t']) == 2            # img + mm, txt + mm
    want, _, _ = oracle.conw_aggregate([v.cpu() for v in captured['i']], captured['g_txt'].cpu(), literal=False)
    np.testing.assert_allclose(captured['agg_i'].cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-6)
    sc = algo.best_scores['test']
    assert 0.0 <= sc['i2t']['recall_1'] <= 100.0 and np.isfinite(sc['rsum'])
    for t in algo.total_local_trainers:
        assert t.last_contrast_loss is not None and torch.isfinite(t.last_contrast_loss)
    assert all(torch.isfinite(p).all() for p in algo.engine.model.parameters())


def test_config4_vit_bertlarge_style_model_steps(dev):
    """BASELINE.json configs[4] in miniature: ViT image trunk + BERT text trunk, d = 768-style head, one server step
    plus a client-style inter + intra contrast step (weight 0.5) on the server features."""
    from creamfl_amd.algorithms.contrast import client_contrast_loss
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(3)
    cfg = _small_cfg(dim=96, cnn='vit_tiny_16')
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(16, dev, seed=9, img=64)
    l0, _ = eng.train_step(b[0], b[1], None, b[3])
    l1, _ = eng.train_step(b[0], b[1], None, b[3])
    assert torch.isfinite(l0) and torch.isfinite(l1)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = eng.model(b[0].contiguous(memory_format=torch.channels_last), b[1], None, b[3])
    f = out['image_features']
    assert f.shape == (16, 96) and f.dtype == torch.float32
    np.testing.assert_allclose(f.detach().norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    gen = torch.Generator().manual_seed(1)
    G = torch.nn.functional.normalize(torch.randn(700, 96, generator=gen), dim=-1).to(dev)
    loss, li, lm = client_contrast_loss(f, G, G.flip(0), list(range(16)), f.detach().roll(1, 0), interintra_weight=0.5)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in eng.model.img_enc.parameters() if p.grad is not None)


def test_training_overfits_a_fixed_batch(dev):
    """End-to-end sanity of every gradient path in the mixed-precision regime of the bench (bf16 trunks with fused
    BN kernels, bf16 trunk weights + fp32 masters, HIP head / loss / clip / AdamP): repeated steps on one fixed batch
    must drive the soft-contrastive loss down and the matched-pair similarity up."""
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(4)
    cfg = _small_cfg(dim=64, cnn='resnet18')
    cfg.optimizer.learning_rate = 5e-4
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    assert eng.model.img_enc.cnn.conv1.weight.dtype == torch.bfloat16          # O2-style trunk weights
    assert eng.model.img_enc.cnn.bn1.weight.dtype == torch.float32 and eng.model.img_enc.fc.weight.dtype == torch.float32
    eng.model.train()
    for m in eng.model.modules():                     # no dropout noise in the trend
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    b = coco_batch(32, dev, seed=11, img=96)
    images = b[0].contiguous(memory_format=torch.channels_last)
    losses = []
    for it in range(80):
        loss, _ = eng.train_step(images, b[1], None, b[3])
        losses.append(loss.item())
    assert np.isfinite(losses).all()
    first, last = np.mean(losses[:5]), np.mean(losses[-5:])
    assert last < 0.25 * first, (first, last)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        out = eng.model(images, b[1], None, b[3])
    sim = out['image_features'] @ out['caption_features'].T
    diag = sim.diag().mean().item()
    off = (sim.sum() - sim.diag().sum()).item() / (32 * 31)
    assert diag > off + 0.2, (diag, off)
    # the fp32 masters and their bf16 shadows stay in sync
    p = eng.model.img_enc.cnn.layer1[0].conv1.weight
    assert torch.equal(p.detach(), eng.optimizer.state[p]['master'].to(torch.bfloat16))


def test_every_parameter_steps_every_step(dev):
    """Regression guard at the engine level (round 3): with the first, address-keyed version of the fused gradient join a
    downsample branch lost its gradient in about one step in nine -- visible as parameters whose `grad` was None at the
    optimizer step (their AdamP step count lags) and as an optimizer that keeps rebuilding its launch plan.  Twenty bf16 steps
    of a bottleneck trunk in one allocator state: the set of parameters that step never changes, all step counts are equal,
    and the optimizer builds its plan exactly once."""
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.algorithms.optimizers import AdamP
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(5)
    cfg = _small_cfg(cnn='resnet50')
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(8, dev, seed=3, bert=True, img=64)
    builds = [0]
    orig = AdamP._upload_meta
    keys = set()
    orig_plan = AdamP._plan

    def counting_plan(self, gi, params, clip_ids):
        plan = orig_plan(self, gi, params, clip_ids)
        keys.add((gi, plan['key']))
        return plan
    AdamP._plan = counting_plan
    try:
        nsteps = 20
        for _ in range(nsteps):
            eng.train_step(b[0], b[1], None, b[3])
        torch.cuda.synchronize()
    finally:
        AdamP._plan = orig_plan
    assert len(keys) == len(eng.optimizer.param_groups), 'the optimizer saw %d different parameter sets' % len(keys)
    steps = {n: eng.optimizer.state[p]['step'] for n, p in eng.model.named_parameters() if p in eng.optimizer.state}
    lag = {n: s for n, s in steps.items() if s != nsteps}
    assert not lag, sorted(lag.items())[:6]
    assert any('downsample' in n for n in steps), 'the trunk under test has no downsample branch'
    del builds, orig


def test_device_prefetcher_on_the_gpu(dev):
    """utils/prefetch.py: batches staged by the copy thread one ahead arrive bit-identical and in order (pageable and pinned
    sources), non-tensor items pass through, a consumer that breaks early does not hang, a loader error reaches the consumer."""
    from creamfl_amd.utils.prefetch import DevicePrefetcher
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    loader = SyntheticCocoLoader(40, 8, seed=5, img=32)
    want = list(loader)
    got = list(DevicePrefetcher(loader, dev))
    assert len(got) == len(want) == 5
    for g, w in zip(got, want):
        assert g[0].is_cuda and g[1].is_cuda and g[3].is_cuda and g[2] is None and g[6] == w[6]
        assert torch.equal(g[0].cpu(), w[0]) and torch.equal(g[1].cpu(), w[1]) and torch.equal(g[3].cpu(), w[3])
    pinned = [tuple(t.pin_memory() if torch.is_tensor(t) else t for t in b) for b in want]
    for g, w in zip(DevicePrefetcher(pinned, dev, depth=1), want):
        assert torch.equal(g[0].cpu(), w[0])
    for i, g in enumerate(DevicePrefetcher(loader, dev)):
        if i == 1:
            break                                             # the copy thread must notice and stop

    def broken():
        yield want[0]
        raise ValueError('loader failed')
    with pytest.raises(ValueError, match='loader failed'):
        list(DevicePrefetcher(broken(), dev))


# ------------------------------------------------------------------------------------------ the bench's own code path, whole model
class _Knobs:
    """Every measurement switch of DESIGN 6.1 flipped to the UNFUSED / single-stream / library form inside this process (they are
    module attributes read at call time), restored on exit."""

    def __init__(self, off):
        self.off = off

    def __enter__(self):
        from creamfl_amd import _lib, ops, streams
        from creamfl_amd.networks import backbones as bb
        from creamfl_amd.networks.models import pcme as pc
        self.mods = (ops, bb, pc, streams)
        self.saved = (ops._NO_JOIN_FUSE, ops.CONV_STATS[0], ops._NO_STEM_TAIL, ops._NO_STEM_S2D, ops._NO_FWD_DGRAD,
                      bb._NO_CONV_SPLIT, bb._NO_SIDE_WGRAD, bb._NO_ATTN_SMALL, pc._NO_TWO_STREAM)
        self.bres = None
        if self.off:
            ops._NO_JOIN_FUSE, ops.CONV_STATS[0], ops._NO_STEM_TAIL, ops._NO_STEM_S2D, ops._NO_FWD_DGRAD = True, False, True, True, True
            bb._NO_CONV_SPLIT, bb._NO_SIDE_WGRAD, bb._NO_ATTN_SMALL, pc._NO_TWO_STREAM = True, True, True, True
            self.bres = _lib.load().cfl_gemm_bf16_bres_min_m(1 << 30)
        return self

    def __exit__(self, *exc):
        from creamfl_amd import _lib
        ops, bb, pc, _ = self.mods
        (ops._NO_JOIN_FUSE, ops.CONV_STATS[0], ops._NO_STEM_TAIL, ops._NO_STEM_S2D, ops._NO_FWD_DGRAD,
         bb._NO_CONV_SPLIT, bb._NO_SIDE_WGRAD, bb._NO_ATTN_SMALL, pc._NO_TWO_STREAM) = self.saved
        if self.bres is not None:
            _lib.load().cfl_gemm_bf16_bres_min_m(self.bres)


def _bench_path_run(dev, unfused, steps, batch, state=None, fp32=False, lr=5e-6):
    """`steps` server steps of a ResNet-50 + BERT-mini PCME at bf16 (to_half: bf16 trunk weights, fp32 masters) on one batch.
    Returns (losses, {name: gradient of step 1 as fp32 CPU}, names without a gradient per step, initial state_dict).
    The model is CONDITIONED like a trained one: the last BatchNorm scale of every residual block starts at 0.25 instead of 1.
    At the plain random initialisation the gradient norm grows ~370x from layer4 to the stem (measured: the stem convolution
    carried 91 % of the gradient norm) and every rounding difference grows with it -- two IDENTICAL bf16 runs then agree only to
    cosine 0.995 / relative 0.10 on the low layers and the fp32 run to cosine ~0 (ReLU patterns decorrelate), which says
    nothing about the kernels."""
    from creamfl_amd import runtime
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.networks.backbones import BasicBlock, Bottleneck
    runtime.configure()
    # the library convolutions in immediate mode: PyTorch's benchmark mode re-times every MIOpen solver of every new problem
    # (~60 s per model variant here, 4 variants); which library kernel runs is not what this test is about
    with _Knobs(unfused), torch.backends.cudnn.flags(enabled=True, benchmark=False):
        torch.manual_seed(11)
        cfg = _small_cfg(dim=128, cnn='resnet50')
        cfg.optimizer.learning_rate = lr
        eng = TrainerEngine(device=dev)
        eng.create(cfg, {'<pad>': 0}, None, False)
        for m in eng.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, Bottleneck):
                torch.nn.init.constant_(m.bn3.weight, 0.25)
            elif isinstance(m, BasicBlock):
                torch.nn.init.constant_(m.bn2.weight, 0.25)
        if state is not None:
            eng.model.load_state_dict(state)
        state0 = copy.deepcopy(eng.model.state_dict())
        eng.model_to_device()
        if not fp32:
            eng.to_half()
        eng.model.train()
        losses, grads1, missing = [], None, []
        for s in range(steps):
            loss, _ = eng.train_step(*batch)
            torch.cuda.synchronize()
            losses.append(float(loss.detach()))
            missing.append([n for n, p in eng.model.named_parameters() if p.requires_grad and p.grad is None])
            if s == 0:
                grads1 = {n: p.grad.detach().float().cpu() for n, p in eng.model.named_parameters() if p.grad is not None}
        return losses, grads1, missing, state0


def _grad_agreement(got, want):
    """per-parameter (relative L2 error, cosine) of two gradient dicts"""
    out = {}
    for n, w in want.items():
        g = got[n].double().flatten()
        w = w.double().flatten()
        nw = float(w.norm())
        rel = float((g - w).norm()) / (nw + 1e-30)
        cos = float(torch.dot(g, w) / (g.norm() * w.norm() + 1e-30))
        out[n] = (rel, cos, nw)
    return out


def _group_cos(g, h, pred):
    names = [n for n in sorted(g) if pred(n)]
    a = torch.cat([g[n].double().flatten() for n in names])
    b = torch.cat([h[n].double().flatten() for n in names])
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def test_bench_code_path_whole_model_fused_vs_unfused(dev):
    """VERDICT r3 #3 / missing #4.  The code path the BENCH times -- bf16 channels_last trunks with every fusion on (fused
    BatchNorm + join GEMM, B-resident data gradients, conv-epilogue statistics, space-to-depth stem, fused stem tail, k x k data
    gradients on forward kernels, short-caption attention, text tower and weight gradients on side streams) -- whole model
    (ResNet-50 + BERT-mini, batch 16, six steps on one batch; step semantics: retrieval_trainer.py:192-214) against
      (a) ITSELF, run again from the same state: a race between the step's streams would show as run-to-run noise;
      (b) the same model with every CFL_NO_* knob off (library convolutions and data gradients, single stream);
      (c) fp32 trunks (one step).
    What can be asked of such a comparison was MEASURED first (docs/history/tools/guard_debug.py, profiles/r4_guard_calibration.txt): the
    library's bf16 weight-gradient / backward-data kernels are not run-to-run reproducible -- two identical UNFUSED runs agree
    per parameter only to cosine 0.68 in the low layers (two identical fused runs, whose data gradients are the deterministic
    hand-written GEMMs: 0.9965), and the trunk as a whole to ~0.8 -- so per-parameter cosine 0.999 is not a property even of the
    reference path.  The bounds below sit between the measured noise and what a real defect produces:
      * no parameter is EVER without a gradient (the address-keyed join of commit 27c5592 dropped a downsample branch's);
      * loss trajectories within 1e-2 (measured 1e-3);
      * head and text-tower gradients: per parameter relative L2 <= 0.25, cosine >= 0.98 (measured worst 0.12 / 0.9925 at the PIE
        attention weights, which see the trunk's activations), all of them together cosine >= 0.9999;
      * every major trunk parameter: gradient NORM within [0.8, 1.25] of the reference run's (measured 0.92 .. 1.05) and cosine
        >= 0.4 -- a dropped residual branch, a doubled gradient or a sign error moves these far outside; whole-trunk cosine >= 0.65;
      * fused vs fused: per parameter cosine >= 0.99, whole model >= 0.9999."""
    from creamfl_amd.utils.synthetic import coco_batch
    b = coco_batch(16, dev, seed=21, bert=True)
    batch = (b[0], b[1], None, b[3])
    lf, gf, mf, state0 = _bench_path_run(dev, False, 6, batch)
    lf2, gf2, mf2, _ = _bench_path_run(dev, False, 6, batch, state=state0)
    lu, gu, mu, _ = _bench_path_run(dev, True, 6, batch, state=state0)
    for missing in (mf, mf2, mu):
        assert all(not m for m in missing), [m for m in missing if m]
    assert set(gf) == set(gu) == set(gf2)
    assert lf[-1] < 0.9 * lf[0]                                       # the model did train
    np.testing.assert_allclose(lf2, lf, rtol=1e-2)
    np.testing.assert_allclose(lu, lf, rtol=1e-2)
    is_trunk = lambda n: n.startswith('img_enc.cnn.')                 # noqa: E731
    total = sum(float(v.double().norm()) ** 2 for v in gf.values()) ** 0.5
    major = [n for n, v in gf.items() if float(v.double().norm()) > 1e-3 * total]
    assert len(major) > 100
    # (a) the fused path against itself
    again = _grad_agreement(gf2, gf)
    assert min(again[n][1] for n in major) >= 0.99, sorted((again[n][1], n) for n in major)[:4]
    assert _group_cos(gf2, gf, lambda n: True) >= 0.9999
    # (b) against the unfused path
    agree = _grad_agreement(gf, gu)
    for n in major:
        rel, cos, _ = agree[n]
        if is_trunk(n):
            ratio = float(gf[n].double().norm() / (gu[n].double().norm() + 1e-30))
            assert 0.8 <= ratio <= 1.25 and cos >= 0.4, (n, ratio, cos)
        else:
            # (the PIE head's attention weights see the trunk's [N, 49, Cd] activations, whose bf16 rounding differs between the
            # two convolution paths: measured relative 0.12 / cosine 0.9925 there, 1e-2 / 0.9999 for the other heads)
            assert rel <= 0.25 and cos >= 0.98, (n, rel, cos)
    assert _group_cos(gf, gu, is_trunk) >= 0.65
    assert _group_cos(gf, gu, lambda n: not is_trunk(n)) >= 0.9999
    # (c) once against full-precision trunks (fp32 weights and activations; the fused bf16 kernels are not active there)
    l32, g32, m32, _ = _bench_path_run(dev, False, 1, batch, state=state0, fp32=True)
    assert not m32[0]
    np.testing.assert_allclose(lf[0], l32[0], rtol=1e-2)
    a32 = _grad_agreement(gf, g32)
    for n in major:
        if not is_trunk(n):
            assert a32[n][0] <= 0.25 and a32[n][1] >= 0.98, (n, a32[n][:2])
    assert _group_cos(gf, g32, lambda n: True) >= 0.99
    print('bench-path guard: fused vs fused worst cosine %.4f; vs unfused trunk cosine %.3f, worst major cosine %.3f; vs fp32 whole-model '
          'cosine %.4f' % (min(again[n][1] for n in major), _group_cos(gf, gu, is_trunk), min(agree[n][1] for n in major),
                           _group_cos(gf, g32, lambda n: True)))


def test_client_contrast_step_in_a_hip_graph_equals_eager(dev):
    """VERDICT r3 #5.  An image client's contrast loop (ClientTrainer.py:369-429: features, old-model features, inter + intra
    contrast against the global banks, backward, SGD step) replayed from ONE HIP graph after three eager steps
    (creamfl_amd/graphs.py; --client_graph 1, the default) against the same loop run eagerly (--client_graph 0): same
    parameters after 9 steps (library convolutions may reorder sums: 1e-4 of scale), the ragged last batch runs eagerly, and the
    graph was really replayed."""
    from creamfl_amd.algorithms.ClientTrainer import ClientTrainer
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    M, D, bs = 136, 64, 16                                             # 8 full batches + one ragged batch of 8
    loader = SyntheticCocoLoader(M, bs, seed=7, img=64)
    batches = list(loader)
    gen = torch.Generator().manual_seed(3)
    g_img = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    g_txt = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    distill_index = list(range(M))

    def run(graph):
        args = SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1, contrast_local_intra=True, contrast_local_inter=True,
                               interintra_weight=0.5, loss_scale=False, save_client=False, client_graph=graph)
        t = ClientTrainer(args, 'Cifar100', None, None, None, None, None, global_test_set=None, client_id=0, gpuid=str(dev))
        t.train_loader = None
        t.cur_epoch = 0
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):   # (immediate mode: no minute of solver timing per variant)
            t.run(g_img, g_txt, distill_index, batches)
        torch.cuda.synchronize()
        return t, {k: v.detach().float().cpu() for k, v in t.model.state_dict().items()}

    t_eager, sd_eager = run(0)
    t_graph, sd_graph = run(1)
    gs = t_graph.graph_stats                                           # (the graph itself is released at the end of run())
    assert gs['failed'] is None, gs['failed']
    assert gs['calls'] == 9 and gs['replays'] == 5                     # 3 eager warm-up steps, 5 replays, the ragged batch eager
    assert t_eager.graph_stats is None and t_graph._graphed_contrast is None
    assert bool(torch.isfinite(t_graph.last_contrast_loss))
    moved = 0.0
    for k, v in sd_eager.items():
        if not v.is_floating_point():
            assert torch.equal(v, sd_graph[k]), k                      # BatchNorm batch counters advance inside the graph too
            continue
        scale = float(v.abs().max()) + 1e-12
        assert float((v - sd_graph[k]).abs().max()) <= 1e-4 * scale + 1e-6, k      # (measured 2e-5 of scale: the library's atomics)
    ref = ClientTrainer(SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1), 'Cifar100', None, None, None, None, None,
                        global_test_set=None, client_id=0, gpuid=str(dev)).model.state_dict()
    moved = max(float((sd_graph[k] - ref[k].float().cpu()).abs().max()) for k in sd_graph if sd_graph[k].is_floating_point())
    assert moved > 1e-6                                                # the steps did train


@pytest.mark.gpu
@pytest.mark.parametrize('M,D,bs', [(136, 64, 16), (1088, 256, 128)])      # (the second: 128 x 32 = 4096 embedding indices per step)
def test_text_client_contrast_step_in_a_hip_graph_equals_eager(dev, M, D, bs):
    """A TEXT client's contrast loop replayed from one HIP graph (possible since gru.hip keeps the caption lengths on the device:
    no packed sequences): every batch padded to one caption width, three eager steps, one capture, replays; against the same loop
    run eagerly on the unpadded batches (--client_graph 0).  Same parameters after 9 steps (1e-4 of scale), the ragged last batch
    runs eagerly, the graph was really replayed, and a batch WIDER than the captured width runs eagerly too."""
    from creamfl_amd.algorithms.ClientTrainer import ClientTrainer, caption_graph_width, pad_captions
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    batches = list(SyntheticCocoLoader(M, bs, seed=7, img=8, bert=False))         # 8 full batches + one ragged batch of bs / 2
    widths = {b[1].shape[1] for b in batches}
    assert (len(widths) > 1 or bs > 16) and max(widths) <= 32          # (small batches differ in width;) all fit the captured one
    assert caption_graph_width(max(widths)) == 32 and pad_captions(batches[0][1], 32).shape[1] == 32
    gen = torch.Generator().manual_seed(3)
    g_img = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    g_txt = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)

    def run(graph, data):
        args = SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1, contrast_local_intra=True, contrast_local_inter=True,
                               interintra_weight=0.5, loss_scale=False, save_client=False, client_graph=graph)
        t = ClientTrainer(args, 'AG_NEWS', None, None, None, None, None, global_test_set=None, client_id=0, gpuid=str(dev))
        t.train_loader = None
        t.cur_epoch = 0
        t.run(g_img, g_txt, list(range(M)), data)
        torch.cuda.synchronize()
        return t, {k: v.detach().float().cpu() for k, v in t.model.state_dict().items()}

    t_eager, sd_eager = run(0, batches)
    t_graph, sd_graph = run(1, batches)
    gs = t_graph.graph_stats
    assert gs['failed'] is None, gs['failed']
    assert gs['calls'] == 9 and gs['replays'] == 5                     # 3 eager warm-up steps, 5 replays, the ragged batch eager
    assert t_eager.graph_stats is None
    assert bool(torch.isfinite(t_graph.last_contrast_loss))
    for k, v in sd_eager.items():
        scale = float(v.abs().max()) + 1e-12
        assert float((v - sd_graph[k]).abs().max()) <= 1e-4 * scale + 1e-6, k
    ref = ClientTrainer(SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1), 'AG_NEWS', None, None, None, None, None,
                        global_test_set=None, client_id=0, gpuid=str(dev)).model.state_dict()
    assert max(float((sd_graph[k] - ref[k].float().cpu()).abs().max()) for k in sd_graph) > 1e-6      # the steps did train
    # a batch wider than the captured width: that call is eager, the loop goes on
    wide = list(batches[:8])
    b5 = list(wide[5])
    b5[1] = pad_captions(b5[1], 40)
    wide[5] = tuple(b5)
    t_wide, sd_wide = run(1, wide)
    assert t_wide.graph_stats['failed'] is None and t_wide.graph_stats['calls'] == 8 and t_wide.graph_stats['replays'] == 4
    t_ref, sd_ref = run(0, batches[:8])
    for k, v in sd_ref.items():
        assert float((v - sd_wide[k]).abs().max()) <= 1e-4 * (float(v.abs().max()) + 1e-12) + 1e-6, k


@pytest.mark.gpu
def test_first_server_step_answers_from_the_find_db(dev):
    """Every process used to spend ~56 s of its first server step letting MIOpen time solvers whose answers the shipped find-db
    already holds (PyTorch's cudnn.benchmark asks for an exhaustive search).  The trunk's convolution calls now use immediate mode
    per call where the problem's key is in the find-db of the process (ops._fdb_covered) and keep the timed search elsewhere:
    at BASELINE configs[1] every problem is covered, and the first step takes seconds."""
    import time
    from creamfl_amd import ops
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    if not ops._FDB['on']:
        pytest.skip('CFL_MIOPEN_AUTO=0')
    torch.manual_seed(7)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(256, dev, seed=7, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)
    saved, ops._FDB['known'] = ops._FDB['known'], {}               # this step's problems only (the decision is re-derived from the db)
    try:
        torch.cuda.synchronize()
        t0 = time.time()
        loss, _ = eng.train_step(images, b[1], b[2], b[3])
        torch.cuda.synchronize()
        dt = time.time() - t0
        new = dict(ops._FDB['known'])
    finally:
        saved.update(ops._FDB['known'])
        ops._FDB['known'] = saved
    assert torch.isfinite(loss)
    # (37 library problems since round 6: the 3 x 3 / stride 1 and the 1 x 1 weight gradients up to 28 x 28 run on csrc/wgrad3x3.hip /
    # wgrad1x1.hip and never reach MIOpen; 46 before)
    assert len(new) >= 30 and all(new.values()), [k for k, v in new.items() if not v]
    assert dt < 30.0, dt
    assert torch.backends.cudnn.benchmark is True                    # the per-call switch leaves the process setting alone


@pytest.mark.gpu
def test_bench_forward_flops_counts_both_towers(dev):
    """`mfu` in the bench line counted nn.Linear MODULES only, and the fused BERT path calls F.linear on the
    weights -- the whole text tower was missing (12.2 instead of ~15 TFLOP per step).  bench.forward_flops now wraps F.linear:
    the text tower's count must be the analytic 2 * tokens * (4 H^2 + 2 H I) per layer (+ attention), less what the [CLS]-only
    last layer saves, and the image tower's the ResNet's convolutions + fc + PIE."""
    import bench
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.networks.backbones import BERT_CONFIGS
    from creamfl_amd.utils.synthetic import coco_batch
    cfg = _small_cfg(dim=64, cnn='resnet18', not_bert=False)
    cfg.model.bert_name = 'bert-mini'
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    N = 16
    b = coco_batch(N, dev, seed=5, bert=True, img=64)
    images = b[0].contiguous(memory_format=torch.channels_last)
    f = bench.forward_flops(eng, images, b[1], b[2], b[3])
    c = BERT_CONFIGS['bert-mini']
    H, I, layers = c['hidden_size'], c['intermediate_size'], c['num_hidden_layers']
    L = int(b[1].shape[1])
    # round 6: the tower runs on the batch's real tokens (coco_batch hands the lengths over on the host): T of the N L positions
    T = sum(b[3]._cfl_host_lens)
    assert 0.4 * N * L < T < N * L
    full = layers * (2 * T * (4 * H * H + 2 * H * I) + 4 * sum(n * n for n in b[3]._cfl_host_lens) * H) + 2 * N * H * 64
    assert 0.70 * full <= f['text'] <= 1.001 * full, (f, full)          # the last layer runs for the [CLS] row only
    assert f['text'] >= (layers - 1) / layers * 0.98 * full - 4 * N * L * L * H
    # ResNet-18 at 64 x 64: 1.814 GMAC at 224 x 224 scales with the pixel count (fc + PIE on top)
    conv = 2 * 1.814e9 * (64 * 64) / (224 * 224) * N
    assert 0.9 * conv <= f['image'] <= 1.25 * conv, (f, conv)
    assert f['total'] == f['image'] + f['text']


@pytest.mark.gpu
def test_training_outcome_bf16_fused_equals_fp32(dev):
    """VERDICT r4 missing #4 / next #3b: does the bf16 path TRAIN like the fp32 path?  A synthetic retrieval task with signal
    (tests/learnable_task.py: 200 identity prototypes <-> caption signatures; a model that learns it retrieves ~all of them, an
    untrained one 0.5 %), the same PCME (ResNet-18 + BERT-mini, d = 64) trained by `TrainerEngine.train_step`
    (retrieval_trainer.py:185-214) from ONE initial state -- once with bf16 trunks and every fusion of the bench's code path on,
    once with fp32 trunks -- then `COCOEvaluator.evaluate` (eval_coco.py:392-448) on held-out samples (fresh noise, fresh filler
    words) of the 200 identities x 5 captions.  Both runs must have learned the task and agree in retrieval quality.  (Settings
    calibrated with tools/train_outcome_probe.py and, for learnability, the CPU oracle port: the loss leaves its plateau at
    ~step 100 and R@1 passes 95 % by step 400.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from learnable_task import LearnableTask
    from train_outcome_probe import train_and_eval
    o = OUTCOME
    task = LearnableTask(n_id=o['n_id'], img=64, seed=0, noise=o['noise'], device=dev)
    with torch.backends.cudnn.flags(enabled=True, benchmark=False):
        bf16, state = train_and_eval(task, o['steps'], o['batch'], o['lr'], False, None, dev, n_eval=o['n_id'])
    with torch.backends.cudnn.flags(enabled=True, benchmark=True):
        # (fp32 convolutions in immediate mode run on fallback kernels, ~0.4 s per step here: the timed search pays for itself)
        fp32, _ = train_and_eval(task, o['steps'], o['batch'], o['lr'], True, state, dev, n_eval=o['n_id'])
    for run in (bf16, fp32):
        assert run['losses'][-1] < 0.25 * run['losses'][0], run                     # the loss left its plateau
        assert run['i2t_r1'] >= o['min_r1'] and run['t2i_r1'] >= o['min_r1'], run    # learned (chance: 0.5)
    for k in ('i2t_r1', 't2i_r1', 'fold_i2t_r1', 'fold_t2i_r1'):
        assert abs(bf16[k] - fp32[k]) <= o['band'], (k, bf16, fp32)


# the training-outcome test: task size, steps / batch / learning rate, and its acceptance (R@1 points), calibrated with
# tools/train_outcome_probe.py on an MI355X (profiles/r5_train_outcome.jsonl)
OUTCOME = {'n_id': 200, 'noise': 0.3, 'steps': 400, 'batch': 32, 'lr': 2e-4, 'min_r1': 90.0, 'band': 6.0}
# measured on four leases (i2t / t2i): bf16 fused 98.0 / 98.2, 99.5 / 99.5, 99.5 / 99.5, 96.5 / 96.1; fp32 99.0 / 99.0, 99.0 / 99.5,
# 99.0 / 99.5, 99.0 / 100.0 -- the bf16 run's spread from run to run (library atomics in its weight gradients) is what sets the band


# the tighter form (round 6): an AMBIGUOUS task (25 % of all captions, training and held-out, carry another identity's signature:
# tests/learnable_task.py `caption_swap`), so a converged model ends near R@1 = 75 % in both directions instead of 99 %; three
# model seeds per precision in one process.  Calibration (tools/train_outcome_probe.py --n-id 200 --caption-swap 0.25 --seeds 3,
# profiles/r6_outcome_calibration_ambiguous.jsonl, i2t / t2i per seed): 700 steps: bf16 77.0 / 74.9, 78.0 / 74.4, 73.5 / 75.0; fp32
# 76.5 / 74.6, 78.5 / 74.2, 69.0 / 73.1 -> means 76.2 / 74.8 vs 74.7 / 74.0.  (400 steps is still on the slope: 43 ... 73.)  t2i has
# 1000 queries per run and is the asserted 2-point quantity.  i2t has 200 queries per run (one query = 0.5 points, +- 3 points of
# sampling noise per run, +- 2.5 on the difference of two three-run means): over three recorded runs of this protocol the bf16 mean sat
# 1.5, 2.8 and 5.2 points ABOVE the fp32 mean (79.5 vs 74.3 on the last: t2i 73.6 vs 74.4 in the same run) -- so the i2t check is
# one-sided at 4 points (bf16 must not learn the clean pairs LESS sharply) and two-sided at 8.
OUTCOME3 = {'n_id': 200, 'noise': 0.3, 'caption_swap': 0.25, 'steps': 700, 'batch': 32, 'lr': 2e-4, 'seeds': 3,
            'band_t2i': 2.0, 'band_i2t': 4.0, 'band_i2t_two_sided': 8.0, 'floor': 50.0, 'ceiling': 85.0}    # (floor: a straggler has been seen at 60.6; chance is 0.5)


@pytest.mark.gpu
def test_training_outcome_ambiguous_task_three_seeds(dev):
    """"trains the same", not "trains".  Same protocol as the test above (one initial state per seed,
    bf16 fused trunks vs fp32 trunks, `TrainerEngine.train_step`, `COCOEvaluator.evaluate` on held-out samples:
    retrieval_trainer.py:185-214, eval_coco.py:392-448) on a task whose ceiling is set by the data at R@1 ~ 75 %; the MEDIANS over
    three seeds must agree to 2 points (t2i, 1000 queries per run); i2t (200 queries per run): bf16 at most 4 points below, 8 apart; every run must sit
    in the task's band -- a path that learns the clean pairs less sharply shows up here, it cannot at 99 %."""
    import json
    import statistics
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from learnable_task import LearnableTask
    from train_outcome_probe import train_and_eval
    o = OUTCOME3
    task = LearnableTask(n_id=o['n_id'], img=64, seed=0, noise=o['noise'], device=dev, caption_swap=o['caption_swap'])
    runs = {'bf16': [], 'fp32': []}
    for seed in range(3, 3 + o['seeds']):
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            a, state = train_and_eval(task, o['steps'], o['batch'], o['lr'], False, None, dev, n_eval=o['n_id'], seed=seed)
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            b, _ = train_and_eval(task, o['steps'], o['batch'], o['lr'], True, state, dev, n_eval=o['n_id'], seed=seed)
        runs['bf16'].append(a)
        runs['fp32'].append(b)
    rep = {}
    for k in ('i2t_r1', 't2i_r1'):
        for prec in ('bf16', 'fp32'):
            v = [r[k] for r in runs[prec]]
            rep[f'{prec}_{k}'] = {'runs': v, 'mean': round(statistics.mean(v), 2), 'median': round(statistics.median(v), 2),
                                  'spread': round(max(v) - min(v), 2)}
    print('training outcome, ambiguous task:', json.dumps(rep))
    for prec in ('bf16', 'fp32'):
        for r in runs[prec]:
            assert r['losses'][-1] < 0.25 * r['losses'][0], r
            assert o['floor'] <= r['i2t_r1'] <= o['ceiling'] and o['floor'] <= r['t2i_r1'] <= o['ceiling'], (prec, r)
    # The compared statistic is the MEDIAN of the three seeds.  At this budget a run of EITHER precision can still be on the slope (the
    # step at which a run leaves the plateau is chaotic): the calibration has an fp32 seed at 69.0 i2t beside 76.5 / 78.5, the fifth
    # recorded run of this test a bf16 seed at 60.6 t2i beside 74.6 / 74.3 (fp32 74.9 / 75.4 / 74.3) -- one straggler moves a three-run
    # mean by 4-5 points and says nothing about where the precision converges.  A straggler is allowed once per precision, not twice.
    for k in ('i2t_r1', 't2i_r1'):
        for prec in ('bf16', 'fp32'):
            r = rep[f'{prec}_{k}']
            assert sum(v < r['median'] - 8.0 for v in r['runs']) <= 1, rep
    assert abs(rep['bf16_t2i_r1']['median'] - rep['fp32_t2i_r1']['median']) <= o['band_t2i'], rep
    assert rep['fp32_i2t_r1']['median'] - rep['bf16_i2t_r1']['median'] <= o['band_i2t'], rep
    assert abs(rep['bf16_i2t_r1']['median'] - rep['fp32_i2t_r1']['median']) <= o['band_i2t_two_sided'], rep


@pytest.mark.gpu
def test_image_client_layouts_train_the_same(dev):
    """Round 5: the clients' image encoders run channels_last by default (`--client_channels_last 1`: the reference's fp32
    arithmetic, ClientTrainer.py:369-429, on the library's NHWC kernels: 28.6 -> 23.9 ms per contrast step at B = 128) and can run
    under bf16 autocast as an opt-in (`--client_bf16 1`: 8.5 ms, BELOW the reference's client precision).  The same client, the same
    batches, 8 eager contrast steps: channels_last must end where NCHW ends to fp32 / algorithm-choice noise; bf16 stays within
    bf16 noise of it and is never the default."""
    from creamfl_amd import flags
    from creamfl_amd.algorithms.ClientTrainer import ClientTrainer
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    assert flags.BUILD_FLAGS['client_channels_last'][0]['default'] == 1 and flags.BUILD_FLAGS['client_bf16'][0]['default'] == 0
    M, D, bs = 128, 64, 16
    batches = list(SyntheticCocoLoader(M, bs, seed=9, img=64))
    gen = torch.Generator().manual_seed(5)
    g_img = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    g_txt = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)

    def run(cl, bf16, x3=0):
        from creamfl_amd import ops
        args = SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1, contrast_local_intra=True, contrast_local_inter=True,
                               interintra_weight=0.5, loss_scale=False, save_client=False, client_graph=0, client_channels_last=cl,
                               client_bf16=bf16, client_conv_x3=x3)
        was = ops.X3CONV[0]
        ops.X3CONV[0] = False
        taken = ops.X3CONV_TAKEN[0]
        try:
            t = ClientTrainer(args, 'Cifar100', None, None, None, None, None, global_test_set=None, client_id=0, gpuid=str(dev))
            assert ops.X3CONV[0] == bool(x3)
            t.train_loader = None
            t.cur_epoch = 0
            with torch.backends.cudnn.flags(enabled=True, benchmark=False):
                t.run(g_img, g_txt, list(range(M)), batches)
                vec, _ = t.generate_logits(batches)
            torch.cuda.synchronize()
            assert (ops.X3CONV_TAKEN[0] > taken) == bool(x3)
        finally:
            ops.X3CONV[0] = was
        first = next(t.model.parameters())
        assert first.is_contiguous(memory_format=torch.channels_last) == bool(cl or bf16) or first.dim() != 4
        sd = {k: v.detach().float().cpu() for k, v in t.model.state_dict().items() if v.is_floating_point()}
        return sd, float(t.last_contrast_loss), vec['img'].float().cpu()
    ref, loss_ref, rep_ref = run(0, 0)
    cl, loss_cl, rep_cl = run(1, 0)
    bf, loss_bf, rep_bf = run(0, 1)
    assert abs(loss_cl - loss_ref) <= 1e-4 * abs(loss_ref) + 1e-5, (loss_cl, loss_ref)
    assert float((rep_cl - rep_ref).abs().max()) <= 1e-3                       # unit-norm representations (measured 2.8e-4: the
    # library runs other fp32 algorithms -- Winograd in NCHW, implicit GEMM in NHWC)
    for k, v in ref.items():
        scale = float(v.abs().max()) + 1e-12
        assert float((cl[k] - v).abs().max()) <= 1e-3 * scale + 1e-6, k
    assert abs(loss_bf - loss_ref) <= 3e-2 * abs(loss_ref) and np.isfinite(loss_bf), (loss_bf, loss_ref)
    assert float((rep_bf - rep_ref).abs().max()) <= 5e-2
    # round 6: the 3 x 3 / stride-1 convolutions on csrc/conv3x3_x3.hip (3 x bf16-split products, 16 mantissa bits per operand):
    # held to the SAME bounds as the layout change above -- it is an fp32-class path, not a reduced-precision one
    x3, loss_x3, rep_x3 = run(1, 0, x3=1)
    assert abs(loss_x3 - loss_ref) <= 1e-4 * abs(loss_ref) + 1e-5, (loss_x3, loss_ref)
    assert float((rep_x3 - rep_ref).abs().max()) <= 1e-3
    for k, v in ref.items():
        scale = float(v.abs().max()) + 1e-12
        assert float((x3[k] - v).abs().max()) <= 1e-3 * scale + 1e-6, k


@pytest.mark.gpu
def test_mm_client_contrast_step_in_a_hip_graph_equals_eager(dev):
    """The MULTI-MODAL client's contrast loop (MMClientTrainer.py:150-224: both towers, the old model's forward, stacked intra +
    summed inter terms, backward, clip + AdamP) replayed from one HIP graph per round -- possible since the fused AdamP reads its
    step count from the device inside a graph (cfl_adamp_step_counted) -- against the same loop run eagerly (--mm_client_graph 0).
    Two local epochs: the local PCME steps on the client's own pairs (eager, they also update the criterion's scalars) sit
    between the two epochs' replays of ONE graph.  Weights as close to the eager run's as a second eager run is
    (_assert_same_training), same step counts, the ragged last batch eager, the graph really replayed."""
    from creamfl_amd.algorithms.MMClientTrainer import MMClientTrainer
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import SyntheticCocoLoader
    M, D, bs = 136, 64, 16                                             # 8 full batches + one ragged batch of 8
    pub = list(SyntheticCocoLoader(M, bs, seed=7, img=64, bert=False))
    own = list(SyntheticCocoLoader(32, bs, seed=9, img=64, bert=False))
    gen = torch.Generator().manual_seed(3)
    g_img = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    g_txt = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)

    def run(graph):
        torch.manual_seed(11)
        args = SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=2, contrast_local_intra=True, contrast_local_inter=True,
                               interintra_weight=0.5, loss_scale=False, save_client=False, client_graph=1, mm_client_graph=graph)
        cfg = default_config(embed_dim=D, cnn_type='resnet18', not_bert=True)
        cfg.train.use_fp16 = False
        msgs = []
        t = MMClientTrainer(args, cfg, SimpleNamespace(log=msgs.append), client=0, device=str(dev), train_loader=own)
        w0 = {k: v.detach().float().cpu().clone() for k, v in t.model.state_dict().items()}
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            t.run(g_img, g_txt, list(range(M)), pub)
        torch.cuda.synchronize()
        osd = t.optimizer.state_dict()
        steps = sorted(set(int(st['step']) for st in osd['state'].values()))
        return t, {k: v.detach().float().cpu() for k, v in t.model.state_dict().items()}, steps, msgs, w0

    t_e, sd_e, steps_e, _, w0 = run(0)
    _, sd_e2, _, _, _ = run(0)
    t_g, sd_g, steps_g, msgs, _ = run(1)
    gs = t_g.graph_stats
    assert gs is not None and gs['failed'] is None, (gs, msgs)
    assert gs['calls'] == 18 and gs['replays'] == 5 + 8                # epoch 1: 3 warm-ups, capture + 4 replays, ragged; epoch 2: 8 + ragged
    assert t_e.graph_stats is None and t_g._graphed_contrast is None
    assert steps_e == steps_g and max(steps_g) == 2 * (2 + 9)          # model: local + contrast steps; the criterion's scalars: 4
    assert bool(torch.isfinite(t_g.last_contrast_loss))
    for k, v in sd_e.items():
        if not v.is_floating_point():
            assert torch.equal(v, sd_g[k]), k                          # BatchNorm batch counters advance inside the graph too
    fl = [k for k, v in sd_e.items() if v.is_floating_point()]
    _assert_same_training({k: w0[k] for k in fl}, {k: sd_e[k] for k in fl}, {k: sd_e2[k] for k in fl}, {k: sd_g[k] for k in fl},
                          'mm client graph')


def _assert_same_training(w0, w_ref, w_ref2, w_new, what):
    """Do two training runs from one initial state end in the same weights, as far as the run itself is reproducible?  AdamP moves a
    weight by ~lr per step whatever the size of its gradient, so the library convolutions' reordered sums (atomics) decorrelate the
    noise-dominated elements of two IDENTICAL eager runs; the yardstick is therefore measured, not assumed: per parameter
    d(a, b) = |a - b| / |a - initial| (distance relative to the distance trained), and the run under test must be as close to
    the reference run as a second reference run is (x 3 + 0.05 per parameter; x 2 + 0.02 over all parameters together).  A capture
    that replays stale inputs, wrong step counts or constant dropout masks is off by O(1) in EVERY parameter."""
    num_n = num_r = den = 0.0
    for k in w_ref:
        moved = float((w_ref[k] - w0[k]).norm())
        if moved == 0.0:
            assert torch.equal(w_new[k], w_ref[k]), (what, k)
            continue
        d_new = float((w_new[k] - w_ref[k]).norm()) / moved
        d_ref = float((w_ref2[k] - w_ref[k]).norm()) / moved
        assert d_new <= 3.0 * d_ref + 0.05, (what, k, d_new, d_ref)
        num_n += float((w_new[k] - w_ref[k]).norm()) ** 2
        num_r += float((w_ref2[k] - w_ref[k]).norm()) ** 2
        den += moved ** 2
    assert den > 0.0, what
    assert (num_n / den) ** 0.5 <= 2.0 * (num_r / den) ** 0.5 + 0.02, (what, (num_n / den) ** 0.5, (num_r / den) ** 0.5)


def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('half', [False, True])
def test_server_contrastive_step_in_a_hip_graph(dev, half):
    """--server_graph 1: TrainerEngine.train (retrieval_trainer.py:192-214) replays its step from ONE HIP graph -- two towers on two
    streams, MCSoftContrastiveLoss, clip + fused AdamP with the step count on the device, captions padded to one width.
    fp32 trunks, dropout off: the weights after 9 steps are as close to the eager loop's as a second eager run is
    (_assert_same_training), same step counts.  bf16 trunks (the bench's code path: fused
    BatchNorm / GEMM / join kernels, fused BERT glue with dropout ON): the capture succeeds, the graph is replayed, everything
    stays finite, and the loss of the replayed steps stays in the eager loop's range."""
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    batches = [coco_batch(16 if i < 8 else 8, 'cpu', seed=70 + i, bert=True, img=64, min_len=18, max_len=24, index0=16 * i)
               for i in range(9)]
    assert max(b[1].shape[1] for b in batches) <= 24

    def run(graph):
        torch.manual_seed(0)
        cfg = default_config(embed_dim=64, cnn_type='resnet18', not_bert=False)
        cfg.model.bert_name = 'bert-mini'
        eng = TrainerEngine(device=dev)
        msgs = []
        eng.set_logger(SimpleNamespace(log=msgs.append, update_tracker=lambda *a, **k: None))
        eng.create(cfg, {'<pad>': 0}, None, False)
        eng.model_to_device()
        if half:
            eng.to_half()
        else:
            _no_dropout(eng.model)
        eng.server_graph = bool(graph)
        losses = []
        orig = eng.train_step

        def spy(*a, **k):
            out = orig(*a, **k)
            losses.append(out[0].detach())
            return out
        if not graph:
            eng.train_step = spy
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            eng.train(batches)
        torch.cuda.synchronize()
        sd = eng.optimizer.state_dict()
        steps = sorted(set(int(st['step']) for st in sd['state'].values()))
        named = {k: v.detach().float().cpu() for k, v in eng.model.named_parameters()}
        return eng, named, steps, [float(x) for x in losses], msgs

    def initial():
        torch.manual_seed(0)
        cfg = default_config(embed_dim=64, cnn_type='resnet18', not_bert=False)
        cfg.model.bert_name = 'bert-mini'
        eng = TrainerEngine(device='cpu')
        eng.create(cfg, {'<pad>': 0}, None, False)
        return {k: v.detach().float().clone() for k, v in eng.model.named_parameters()}

    e_e, w_e, steps_e, loss_e, _ = run(0)
    e_g, w_g, steps_g, _, msgs = run(1)
    gs = e_g.graph_stats.get('train')
    assert gs is not None and gs['failed'] is None, (gs, msgs)
    assert gs['calls'] == 9 and gs['replays'] == 5                     # 3 warm-ups, capture + 4 replays, the ragged batch eager
    assert e_e.graph_stats == {} and e_g._graphs == {}
    assert steps_e == steps_g == [9]
    assert all(bool(torch.isfinite(v).all()) for v in w_g.values())
    if half:
        return
    _, w_e2, _, _, _ = run(0)
    _assert_same_training(initial(), w_e, w_e2, w_g, 'server graph')


@pytest.mark.gpu
def test_fused_dropout_masks_follow_the_device_tick(dev):
    """A captured step's dropout seeds are constants of the graph; the masks vary between replays through a device word the step
    increments (cfl_set_dropout_tick): same seed + same tick = same mask, another tick = another mask, tick 0 = the mask the seed
    alone gives."""
    from creamfl_amd import ops
    base = ops.dropout_keep_mask(1234, 0.3, (64, 256), dev).clone()
    tick = ops.dropout_tick(dev)
    was = int(tick)
    try:
        tick.zero_()
        k0 = ops.dropout_keep_mask(1234, 0.3, (64, 256), dev).clone()
        tick.add_(1)
        k1 = ops.dropout_keep_mask(1234, 0.3, (64, 256), dev).clone()
        k1b = ops.dropout_keep_mask(1234, 0.3, (64, 256), dev).clone()
        assert torch.equal(k1, k1b) and not torch.equal(k0, k1)
        assert abs(float(k1.float().mean()) - 0.7) < 0.02
        if was == 0:
            assert torch.equal(base, k0)
    finally:
        tick.fill_(was)


@pytest.mark.gpu
def test_one_communication_round_with_server_graphs(dev):
    """The round of test_one_communication_round with --server_graph 1 and the bf16 server: the global-training steps and the KD
    steps (MMFL.py:346-391) each replay from a graph captured in their phase, the multi-modal client replays its contrast step
    too; step counts of the server's optimizer = steps taken, everything finite."""
    from creamfl_amd.algorithms.MMFL import MMFL
    torch.manual_seed(2)
    M = 144
    args = SimpleNamespace(name='/tmp/creamfl_test', feature_dim=64, pub_data_num=M, not_bert=False, mlp_local=False,
                           server_lr=2e-4, local_epochs=1, comm_rounds=1, num_img_clients=1, num_txt_clients=1,
                           num_mm_clients=1, client_num_per_round=3, agg_method='con_w', contrast_local_intra=True,
                           contrast_local_inter=True, interintra_weight=0.5, loss_scale=False, kd_weight=0.3,
                           disable_distill=False, save_client=False, device=0, cnn_type='resnet18', bert_name='bert-mini',
                           image_size=64, test_pairs=100, quiet=True, save_checkpoints=False, server_graph=1)
    algo = MMFL(args, None)
    algo.config.dataloader.batch_size = 16
    algo.create_model(args)
    algo.load_dataset(args)
    algo.train(0)
    st = algo.engine.graph_stats
    assert set(st) == {'train', 'kd'}, st
    for name in ('train', 'kd'):
        assert st[name]['failed'] is None and st[name]['replays'] >= 4, (name, st[name])
    mm = algo.mm_local_trainers[0].graph_stats
    assert mm is not None and mm['failed'] is None and mm['replays'] >= 4, mm
    sd = algo.engine.optimizer.state_dict()
    steps = sorted(set(int(s['step']) for s in sd['state'].values()))
    assert steps == [9, 18], steps                                     # the criterion's scalars skip the 9 KD steps
    assert all(torch.isfinite(p).all() for p in algo.engine.model.parameters())
    assert np.isfinite(algo.best_scores['test']['rsum'])
