"""GPU: one test (at least) per BASELINE.json config, at the config's own sizes, each checked against the CPU oracle.

  configs[0]  single server round, 2 image + 2 text clients, ResNet-18 / BERT-mini, batch 32
  configs[1]  server-only contrastive step, ResNet-101 + BERT-base, d = 512, batch 256      (the bench workload)
  configs[2]  full CreamFL round shape: con_w over 8 client representations of the 50 000-pair public set, d = 256
              (the 8-rank client sharding / all-gather itself is covered on CPU by tests/test_dist_gloo.py)
  configs[3]  large-batch global contrast: N = 4096 (8 x 512), d = 512 pair loss
  configs[4]  d = 768 inter + intra contrast, interintra_weight 0.5, B = 128 against the 50 000-row bank

Tolerances: losses 1e-4 relative (north star), gradients 1e-3 of the tensor's scale, ranks exact.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _unit(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


def _close(got, want, rtol, atol, msg=''):
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64), rtol=rtol, atol=atol,
                               err_msg=msg)


# ------------------------------------------------------------------------------------------------ configs[0]
def test_config0_round_two_image_two_text_clients_batch32(dev):
    """One communication round: 2 image clients + 2 text clients (no multimodal), ResNet-18 / BERT-mini server, batch
    32, inter + intra contrast, con_w, KD.  The con_w aggregate the round computed is re-derived by the oracle from the
    captured client representations; every client trained; the evaluator produced COCO-style scores."""
    from creamfl_amd.algorithms.MMFL import MMFL
    torch.manual_seed(20)
    M = 64
    # the Namespace src/main.py would build (all 41 reference flags at their defaults), overridden only in sizes, plus the
    # build-defined extras (encoder names, synthetic data sizes)
    from conftest import reference_main_namespace
    args, _ = reference_main_namespace(name='/tmp/creamfl_test_cfg0', feature_dim=64, pub_data_num=M, local_epochs=1, comm_rounds=1,
                                       num_img_clients=2, num_txt_clients=2, num_mm_clients=0, client_num_per_round=4,
                                       contrast_local_intra=True, contrast_local_inter=True, cnn_type='resnet18',
                                       bert_name='bert-mini', image_size=64, test_pairs=100, quiet=True, save_checkpoints=False)
    assert args.agg_method == 'con_w' and args.kd_weight == 0.3 and args.interintra_weight == 0.5 and not args.disable_distill
    algo = MMFL(args, None)
    algo.config.dataloader.batch_size = 32
    algo.config.train.use_fp16 = False
    algo.create_model(args)
    algo.load_dataset(args)
    captured = {}
    orig = algo.aggregation

    def spy(i_vec, t_vec):
        captured.update(i=[v.clone() for v in i_vec], t=[v.clone() for v in t_vec], g_txt=algo.global_txt_feature.clone(),
                        g_img=algo.global_img_feature.clone())
        out = orig(i_vec, t_vec)
        captured.update(agg_i=out[0].clone(), agg_t=out[1].clone())
        return out

    algo.aggregation = spy
    algo.train(0)
    assert len(captured['i']) == 2 and len(captured['t']) == 2
    want_i, _, _ = oracle.conw_aggregate([v.cpu() for v in captured['i']], captured['g_txt'].cpu(), literal=True)
    want_t, _, _ = oracle.conw_aggregate([v.cpu() for v in captured['t']], captured['g_img'].cpu(), literal=True)
    _close(captured['agg_i'].cpu().numpy(), want_i.numpy(), 1e-4, 1e-6)
    _close(captured['agg_t'].cpu().numpy(), want_t.numpy(), 1e-4, 1e-6)
    assert len(algo.total_local_trainers) == 4
    for t in algo.total_local_trainers:
        assert t.last_contrast_loss is not None and torch.isfinite(t.last_contrast_loss)
    sc = algo.best_scores['test']
    assert 0.0 <= sc['i2t']['recall_1'] <= 100.0 and np.isfinite(sc['rsum'])
    assert all(torch.isfinite(p).all() for p in algo.engine.model.parameters())


# ------------------------------------------------------------------------------------------------ configs[1]
def test_config1_server_step_r101_bertbase_d512_b256(dev):
    """The bench workload itself: two server steps of ResNet-101 + BERT-base, d = 512, batch 256, bf16 trunks.  The
    loss the step returns is re-derived by the CPU oracle (fp64 closed form of probemb.py:221-256) from the very
    features the step fed to the criterion; the loss dict carries the reference's keys; the weights moved."""
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(21)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    seen = []
    crit_forward = eng.criterion.forward

    def spy(image_features, caption_features, *a, **k):
        seen.append((image_features.detach().float().cpu(), caption_features.detach().float().cpu(),
                     float(eng.criterion.negative_scale.detach()), float(eng.criterion.shift.detach())))
        return crit_forward(image_features, caption_features, *a, **k)

    eng.criterion.forward = spy
    b = coco_batch(256, dev, seed=22, bert=True)
    w0 = eng.model.img_enc.fc.weight.detach().clone()
    for step in range(2):
        loss, ld = eng.train_step(b[0], b[1], b[2], b[3])
        I, T, a, s = seen[-1]
        assert I.shape == (256, 512) and T.shape == (256, 512)
        _close(I.norm(dim=1).numpy(), 1.0, 1e-5, 0)
        cf = oracle.pair_loss_closed_form(I, T, a, s)
        _close(loss.item(), float(cf['loss']), 1e-4, 0, f'step {step}')
        assert list(ld.keys())[:3] == ['i2t_loss', 't2i_loss', 'i2t_pos_loss'] and len(ld) == 11
        _close(ld['loss'], loss.item(), 1e-6, 0)
    assert float((eng.model.img_enc.fc.weight.detach() - w0).abs().max()) > 0
    assert all(torch.isfinite(p).all() for p in eng.model.parameters())


# ------------------------------------------------------------------------------------------------ configs[4], encoders
def test_config4_full_size_vit_b16_bert_large_d768(dev):
    """BASELINE.json configs[4] at its OWN encoder sizes: ViT-B/16 (12 x 768, 12 heads) + BERT-large (24 x 1024, 16 heads),
    d = 768, batch 64, bf16 trunks: two server steps -- finite, the returned loss re-derived by the fp64 oracle from the
    features the step fed to the criterion (1e-4), weights moved -- then a client-style inter + intra step (weight 0.5)
    against a 50 000-row bank on those features (the round-3 column-split bank kernel, D = 768)."""
    from creamfl_amd.algorithms.contrast import client_contrast_loss
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(41)
    cfg = default_config(embed_dim=768, cnn_type='vit_b_16', not_bert=False)
    cfg.model.bert_name = 'bert-large-uncased'
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    n_img = sum(p.numel() for p in eng.model.img_enc.parameters())
    n_txt = sum(p.numel() for p in eng.model.txt_enc.parameters())
    assert 80e6 < n_img < 100e6 and 320e6 < n_txt < 350e6, (n_img, n_txt)        # ViT-B/16 ~86 M, BERT-large ~335 M
    seen = []
    crit_forward = eng.criterion.forward

    def spy(image_features, caption_features, *a, **k):
        seen.append((image_features.detach().float().cpu(), caption_features.detach().float().cpu(),
                     float(eng.criterion.negative_scale.detach()), float(eng.criterion.shift.detach())))
        return crit_forward(image_features, caption_features, *a, **k)

    eng.criterion.forward = spy
    b = coco_batch(64, dev, seed=42, bert=True)
    w0 = eng.model.linear.weight.detach().float().clone()
    for step in range(2):
        loss, ld = eng.train_step(b[0], b[1], b[2], b[3])
        assert torch.isfinite(loss)
        I, T, a, s = seen[-1]
        assert I.shape == (64, 768) and T.shape == (64, 768)
        cf = oracle.pair_loss_closed_form(I, T, a, s)
        _close(loss.item(), float(cf['loss']), 1e-4, 0, f'step {step}')
    assert float((eng.model.linear.weight.detach().float() - w0).abs().max()) > 0
    assert all(torch.isfinite(p).all() for p in eng.model.parameters())
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        out = eng.model(b[0].contiguous(memory_format=torch.channels_last), b[1], b[2], b[3])
    f = out['image_features'].detach().clone().requires_grad_(True)
    gen = torch.Generator().manual_seed(43)
    G = torch.nn.functional.normalize(torch.randn(50000, 768, generator=gen), dim=-1).to(dev)
    Gs = torch.nn.functional.normalize(torch.randn(50000, 768, generator=gen), dim=-1).to(dev)
    idx = torch.randperm(50000, generator=gen)[:64].tolist()
    fo = out['caption_features'].detach()
    loss, li, lm = client_contrast_loss(f, Gs, G, idx, fo, interintra_weight=0.5)
    loss.backward()
    cf = oracle.client_contrast_grads_closed_form(f.detach().cpu(), Gs.cpu(), G.cpu(), idx, fo.cpu())
    _close(li.item(), cf['loss_inter'].item(), 1e-4, 0)
    _close(lm.item(), cf['loss_moon'].item(), 1e-4, 0)
    want = (cf['d_moon'].numpy() + cf['d_inter'].numpy()) * 0.5
    _close(f.grad.cpu().numpy(), want, 1e-3, 1e-4 * np.abs(want).max())


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_conw_eight_clients_public_set_50000(dev):
    """con_w at the reference's hard-coded public-set size (MMFL.py:302: 50 000 rows), d = 256, C = 8 client
    representations: log-prob, softmax-over-clients weights and the aggregate on a random sample of rows against the
    fp64 restatement of MMFL.py:304-314 (the full [M, M] fp64 reference would take minutes on the host)."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(23)
    M, D, C = 50000, 256, 8
    G = _unit(gen, M, D)
    vecs = [torch.nn.functional.normalize(G + (0.3 + 0.2 * c) * _unit(gen, M, D), dim=-1) for c in range(C)]
    Gd = G.to(dev)
    vd = [v.to(dev) for v in vecs]
    lp = torch.stack([ops.conw_logprob(v, Gd) for v in vd], 0)                  # [C, M]
    agg, w = ops.conw_combine(vd, lp, return_weights=True)
    rows = torch.randperm(M, generator=gen)[:96]
    G64 = G.double()
    want_lp = torch.stack([(v[rows].double() * G64[rows]).sum(1) - torch.log(torch.exp(v[rows].double() @ G64.T).sum(1))
                           for v in vecs], 0)                                   # MMFL.py:304-307, no max-subtraction
    want_w = torch.softmax(want_lp, dim=0)                                      # :311
    want_agg = sum(vecs[c][rows].double() * want_w[c][:, None] for c in range(C))   # :312-314
    _close(lp[:, rows.to(dev)].cpu().numpy(), want_lp.numpy(), 1e-5, 1e-5)
    _close(w[:, rows.to(dev)].cpu().numpy(), want_w.numpy(), 1e-4, 1e-6)
    _close(agg[rows.to(dev)].cpu().numpy(), want_agg.numpy(), 1e-4, 1e-6)
    _close(w.sum(0).cpu().numpy(), 1.0, 1e-5, 0)


# ------------------------------------------------------------------------------------------------ configs[3]
def _closed_form_big(I, T, a, b, eps=1e-6):
    """fp64 closed form of probemb.py:48-86,185-256 via the GEMM identity, diagonal distances exact."""
    I64, T64 = I.double(), T.double()
    n = I.shape[0]
    d2 = (I64 * I64).sum(1)[:, None] + (T64 * T64).sum(1)[None, :] - 2.0 * I64 @ T64.T
    d2[torch.arange(n), torch.arange(n)] = ((I64 - T64) ** 2).sum(1)
    d = torch.sqrt(d2.clamp_min(0) + eps)
    s = -a * d + b
    m = -torch.ones(n, n, dtype=torch.float64)
    m.fill_diagonal_(1.0)
    nll = torch.nn.functional.softplus(-2.0 * m * s)
    gg = 4.0 * m * torch.sigmoid(-2.0 * m * s)
    c = a * gg / d
    return ({'loss': 2.0 * nll.sum()},
            {'dI': (I64 * c.sum(1, keepdim=True) - c @ T64).numpy(), 'dT': (T64 * c.sum(0)[:, None] - c.t() @ I64).numpy(),
             'da': (gg * d).sum(), 'db': -gg.sum()})


@pytest.mark.parametrize('a,b', [(15.0, 15.0), (5.0, 3.0)])
def test_config3_pair_loss_global_batch_4096_d512(dev, a, b):
    """The all-gathered global batch of configs[3] (8 ranks x 512 pairs, d = 512) through the pair-loss kernels:
    loss, dI, dT, d(negative_scale), d(shift) against the fp64 closed form."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(24)
    n, d = 4096, 512
    I = _unit(gen, n, d)
    T = torch.nn.functional.normalize(I + 0.5 * _unit(gen, n, d), dim=-1)
    Ig, Tg = I.to(dev).requires_grad_(True), T.to(dev).requires_grad_(True)
    ag = torch.tensor([a], device=dev, requires_grad=True)
    bg = torch.tensor([b], device=dev, requires_grad=True)
    loss, _ = ops.pair_loss(Ig, Tg, ag, bg)
    loss.backward()
    cf, g = _closed_form_big(I, T, a, b)
    _close(loss.item(), float(cf['loss']), 1e-4, 0)
    sc = float(np.abs(g['dI']).max())
    _close(Ig.grad.cpu().numpy(), g['dI'], 1e-3, 2e-4 * sc, 'dI')
    _close(Tg.grad.cpu().numpy(), g['dT'], 1e-3, 2e-4 * sc, 'dT')
    _close(ag.grad.item(), float(g['da']), 1e-3, 0)
    _close(bg.grad.item(), float(g['db']), 1e-3, 0)


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.parametrize('loss_scale', [False, True])
def test_config4_inter_intra_d768_b128_m50000(dev, loss_scale):
    """--contrast_local_inter --contrast_local_intra, interintra_weight 0.5, d = 768, client batch 128 against the
    50 000-row global bank: both loss terms, their combination (ClientTrainer.py:416-419) and dF against the oracle."""
    from creamfl_amd.algorithms.contrast import client_contrast_loss
    gen = torch.Generator().manual_seed(25)
    B, M, D = 128, 50000, 768
    g_img = _unit(gen, M, D)
    g_txt = torch.nn.functional.normalize(g_img + 0.7 * _unit(gen, M, D), dim=-1)
    d_idx = [int(v) for v in torch.randperm(M, generator=gen)[:B]]
    base = g_img[d_idx]
    f = torch.nn.functional.normalize(base + 0.6 * _unit(gen, B, D), dim=-1)
    f_old = torch.nn.functional.normalize(base + 0.6 * _unit(gen, B, D), dim=-1)
    fg = f.to(dev).requires_grad_(True)
    loss, li, lm = client_contrast_loss(fg, g_img.to(dev), g_txt.to(dev), d_idx, f_old.to(dev), interintra_weight=0.5,
                                        loss_scale=loss_scale)
    loss.backward()
    cf = oracle.client_contrast_grads_closed_form(f, g_img, g_txt, d_idx, f_old)
    _close(li.item(), cf['loss_inter'].item(), 1e-4, 0)
    _close(lm.item(), cf['loss_moon'].item(), 1e-4, 0)
    wi, wm = cf['loss_inter'].item(), cf['loss_moon'].item()
    if not loss_scale:
        want, dref = (wm + wi) * 0.5, 0.5 * (cf['d_moon'] + cf['d_inter'])
    else:
        r = wi / wm
        want, dref = (wm + wi / r) * 0.5, 0.5 * (cf['d_moon'] + cf['d_inter'] / r)
    _close(loss.item(), want, 1e-4, 0)
    dref = dref.numpy()
    _close(fg.grad.cpu().numpy(), dref, 1e-3, 1e-4 * np.abs(dref).max())
