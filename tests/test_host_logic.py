"""CPU-only tests of host-side logic that needs no kernel: config, synthetic batch contract, lazy loss dict,
row/client sharding arithmetic, the AdamP oracle against a literal per-parameter transcription."""
import math

import numpy as np
import pytest
import torch

from creamfl_amd import dist as cdist
from creamfl_amd.utils.config import Config, default_config
from creamfl_amd.utils.synthetic import SyntheticCocoLoader, coco_batch


def test_config_access():
    c = default_config(embed_dim=512)
    assert c.model.embed_dim == 512 and c.criterion.get('uniform_lambda', 0) == 0
    c.model.embed_dim = 7
    assert c['model']['embed_dim'] == 7
    with pytest.raises(AttributeError):
        _ = c.nope
    assert isinstance(c.copy(), Config)


def test_synthetic_batch_contract():
    images, captions, words, lens, ann, iid, index = coco_batch(6, bert=False, captions_per_image=5, index0=10)
    assert images.shape == (6, 3, 224, 224) and images.dtype == torch.float32
    assert captions.dtype == torch.int64 and captions.shape[0] == 6 and captions.shape[1] == int(lens.max())
    assert torch.all(lens[:-1] >= lens[1:])                              # sorted by length, descending
    for i, l in enumerate(lens.tolist()):
        assert captions[i, 0] == 1 and captions[i, l - 1] == 2 and torch.all(captions[i, l:] == 0)
    assert index == list(range(10, 16)) and iid == [i // 5 for i in index]
    ld = SyntheticCocoLoader(25, 10, captions_per_image=5)
    assert len(ld) == 3 and ld.dataset.n_images == 5 and len(ld.dataset) == 25
    assert sum(len(b[6]) for b in ld) == 25


def test_lazy_loss_dict_matches_reference_keys():
    from creamfl_amd.ops import LazyLossDict
    out8 = torch.tensor([20.0, 7.0, 3.0, 1.0, 2.0, 0, 0, 0])
    d = LazyLossDict(out8, torch.tensor([15.0]), torch.tensor([14.0]))
    assert list(d) == ['i2t_loss', 't2i_loss', 'i2t_pos_loss', 'i2t_neg_loss', 't2i_pos_loss', 't2i_neg_loss',
                       'uniform_loss', 'vib_loss', 'shift', 'negative_scale', 'loss']
    assert d['loss'] == 20.0 and d['i2t_loss'] == 10.0 and d['t2i_pos_loss'] == 7.0 and d['shift'] == 15.0
    assert dict(d)['negative_scale'] == 14.0 and len(d) == 11


def test_row_and_client_sharding():
    for M, W in [(50000, 8), (1000, 3), (130, 4), (5, 8)]:
        covered = []
        for r in range(W):
            r0, r1 = cdist.row_shard(M, r, W)
            assert r0 % 128 == 0 or r0 == M
            covered += list(range(r0, r1))
        assert covered == list(range(M))
    trainers = list('abcdefghij')
    parts = [cdist.shard_clients(trainers, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == trainers and parts[0] == ['a', 'e', 'i'] and parts[3] == ['d', 'h']


def _literal_adamp_step(p, g, st, lr, betas=(0.9, 0.999), eps=1e-8, wd=0.0, delta=0.1, wd_ratio=0.1):
    """Per-parameter transcription of Algorithm 2 with python control flow (fp64)."""
    b1, b2 = betas
    st['t'] += 1
    st['m'] = b1 * st['m'] + (1 - b1) * g
    st['v'] = b2 * st['v'] + (1 - b2) * g * g
    denom = np.sqrt(st['v']) / math.sqrt(1 - b2 ** st['t']) + eps
    pert = st['m'] / denom
    ratio = 1.0
    if p.ndim > 1:
        for view in ('channel', 'layer'):
            P = p.reshape(p.shape[0], -1) if view == 'channel' else p.reshape(1, -1)
            G = g.reshape(P.shape)
            cos = np.abs((P * G).sum(1)) / (np.maximum(np.linalg.norm(P, axis=1), eps) * np.maximum(np.linalg.norm(G, axis=1), eps))
            if cos.max() < delta / math.sqrt(P.shape[1]):
                Pn = P / (np.linalg.norm(P, axis=1, keepdims=True) + eps)
                Q = pert.reshape(P.shape)
                pert = (Q - Pn * (Pn * Q).sum(1, keepdims=True)).reshape(p.shape)
                ratio = wd_ratio
                break
    p = p * (1 - lr * wd * ratio) if wd > 0 else p
    return p - lr / (1 - b1 ** st['t']) * pert


def test_adamp_oracle_matches_literal_algorithm():
    from oracle.adamp import AdamP
    rng = np.random.default_rng(0)
    shapes = [(6, 4, 3, 3), (6,), (5, 40), (1, 300)]
    ps = [rng.standard_normal(s) * 0.1 for s in shapes]
    tp = [torch.nn.Parameter(torch.tensor(p, dtype=torch.float64)) for p in ps]
    opt = AdamP(tp, lr=1e-2, weight_decay=0.01)
    sts = [{'t': 0, 'm': np.zeros(s), 'v': np.zeros(s)} for s in shapes]
    for it in range(3):
        gs = [rng.standard_normal(s) for s in shapes]
        # orthogonalise some gradients so that both projection branches fire
        gs[0] = (gs[0].reshape(6, -1) - ps[0].reshape(6, -1) * ((gs[0].reshape(6, -1) * ps[0].reshape(6, -1)).sum(1, keepdims=True)
                 / (ps[0].reshape(6, -1) ** 2).sum(1, keepdims=True))).reshape(shapes[0])
        gs[3] = gs[3] - ps[3] * (gs[3] * ps[3]).sum() / (ps[3] ** 2).sum()
        for q, g in zip(tp, gs):
            q.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
        ps = [_literal_adamp_step(p, g, st, 1e-2, wd=0.01) for p, g, st in zip(ps, gs, sts)]
        for q, p in zip(tp, ps):
            np.testing.assert_allclose(q.detach().numpy(), p, rtol=1e-9, atol=1e-12)


def test_adamp_oracle_per_parameter_step_counts():
    """adamp==0.3.0 keeps `state['step']` per parameter: one whose gradient is None in some steps (the criterion's scalars in
    every KD phase) lags behind and is bias-corrected with ITS OWN count.  The oracle against the literal per-parameter
    transcription on a schedule where two of four parameters skip steps."""
    from oracle.adamp import AdamP
    rng = np.random.default_rng(1)
    shapes = [(6, 4, 3, 3), (1,), (5, 40), (7,)]
    ps = [rng.standard_normal(s) * 0.1 for s in shapes]
    tp = [torch.nn.Parameter(torch.tensor(p, dtype=torch.float64)) for p in ps]
    opt = AdamP(tp, lr=1e-2, weight_decay=0.01)
    sts = [{'t': 0, 'm': np.zeros(s), 'v': np.zeros(s)} for s in shapes]
    present = [[1, 1, 1, 1], [1, 0, 1, 0], [1, 0, 0, 1], [1, 1, 1, 1], [1, 0, 1, 1]]      # rows = steps, 0 = grad None
    for mask in present:
        gs = [rng.standard_normal(s) for s in shapes]
        for q, g, on in zip(tp, gs, mask):
            q.grad = torch.tensor(g, dtype=torch.float64) if on else None
        opt.step()
        ps = [_literal_adamp_step(p, g, st, 1e-2, wd=0.01) if on else p for p, g, st, on in zip(ps, gs, sts, mask)]
        for q, p in zip(tp, ps):
            np.testing.assert_allclose(q.detach().numpy(), p, rtol=1e-9, atol=1e-12)
    assert [opt.state[q]['step'] for q in tp] == [5, 2, 4, 4] == [st['t'] for st in sts]


def test_adamp_load_state_without_master_into_bf16_model():
    """an optimizer state with moments but no master (a checkpoint taken from an fp32 model) loaded into a model
    whose trunk weights are bf16 must come out with an fp32 master (KeyError in `_plan` before), moments fp32."""
    from creamfl_amd.algorithms.optimizers import AdamP
    w32 = torch.nn.Parameter(torch.randn(8, 4))
    src = AdamP([w32], lr=1e-3)
    src.state[w32].update(step=3, exp_avg=torch.randn(8, 4), exp_avg_sq=torch.rand(8, 4))
    sd = src.state_dict()
    w16 = torch.nn.Parameter(torch.randn(8, 4).to(torch.bfloat16))
    dst = AdamP([w16], lr=1e-3)
    dst.load_state_dict(sd)
    st = dst.state[w16]
    assert st['step'] == 3 and st['exp_avg'].dtype == torch.float32 and st['exp_avg_sq'].dtype == torch.float32
    assert torch.equal(st['exp_avg'], src.state[w32]['exp_avg'])
    assert st['master'].dtype == torch.float32 and torch.equal(st['master'], w16.detach().float())
    # refresh_masters creates a missing master too, preferring full-precision values when it is given them
    del st['master']
    full = torch.randn(8, 4)
    dst.refresh_masters({w16: full})
    assert torch.equal(dst.state[w16]['master'], full)


def test_bert_cls_only_equals_full_forward_on_cpu():
    """`cls_only=True` evaluates the last layer for [CLS] only: same values and parameter gradients for everything PCME
    consumes (it reads [:, 0, :] only, src/networks/models/pcme.py:44)."""
    import torch
    from creamfl_amd.networks.backbones import BertModel
    torch.manual_seed(0)
    m = BertModel('bert-mini').train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    ids = torch.randint(1, 1000, (3, 9))
    mask = torch.arange(9)[None] < torch.tensor([9, 5, 7])[:, None]
    w = torch.randn(3, 256)
    outs, grads = [], []
    for cls_only in (False, True):
        m.zero_grad(set_to_none=True)
        o = m(ids, attention_mask=mask, cls_only=cls_only)['last_hidden_state'][:, 0]
        (o * w).sum().backward()
        outs.append(o.detach())
        grads.append({n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()})
    assert outs[1].shape == (3, 256)
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=1e-6)
    for n in grads[0]:
        if n.endswith('key.bias'):
            continue                                  # analytically zero (softmax is shift-invariant): rounding noise
        a, b = grads[0][n], grads[1][n]
        if a is None or b is None:
            assert (a is None or float(a.abs().max()) < 1e-6) and (b is None or float(b.abs().max()) < 1e-6), n
        else:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * (float(a.abs().max()) + 1e-3), msg=n)


def test_bnact_counts_batches_on_the_host_and_folds_them_into_the_state():
    """BNAct keeps `num_batches_tracked` increments of the fused path on the host; state_dict / load_state_dict /
    the library path must observe exactly what nn.BatchNorm2d would."""
    import copy
    import torch
    from creamfl_amd.networks.backbones import BNAct
    bn = BNAct(8).train()
    x = torch.randn(4, 8, 3, 3)
    bn(x)
    bn(x)                                   # CPU input -> library path, counts directly
    assert int(bn.state_dict()['num_batches_tracked']) == 2
    bn._nbt_pending = 5                     # what the fused GPU path does: count on the host
    assert int(bn.state_dict()['num_batches_tracked']) == 7 and bn._nbt_pending == 0
    other = copy.deepcopy(bn)
    other._nbt_pending = 3
    other.load_state_dict(bn.state_dict())  # a load replaces the counter, pending increments are dropped
    assert int(other.state_dict()['num_batches_tracked']) == 7
    ref = torch.nn.BatchNorm2d(8)
    ref.load_state_dict(bn.state_dict())    # identical key set / shapes
    assert int(ref.num_batches_tracked) == 7


def test_residual_blocks_pass_pairs_and_match_plain_composition_on_cpu():
    """Bottleneck / BasicBlock hand (conv input, residual input) pairs to each other; on the library path that must be
    the ordinary ResNet block arithmetic."""
    import torch
    import torch.nn.functional as F
    from creamfl_amd.networks.backbones import Bottleneck, first_of
    torch.manual_seed(1)
    blk = Bottleneck(64, 16).eval()
    x = torch.randn(2, 64, 5, 5)
    y = blk(x)
    assert isinstance(y, tuple) and y[0] is y[1]
    y2 = blk((x, x))
    torch.testing.assert_close(first_of(y), first_of(y2))
    o = F.relu(blk.bn1(blk.conv1(x)))
    o = F.relu(blk.bn2(blk.conv2(o)))
    ref = F.relu(blk.bn3(blk.conv3(o)) + x)
    torch.testing.assert_close(first_of(y), ref)


def test_bert_mirror_matches_transformers_on_cpu():
    """The text tower is third-party code in the reference (`transformers.BertModel`, src/networks/models/pcme.py:36-38).
    The mirror in networks/backbones.py loads the library's own state_dict and must reproduce its hidden states and
    parameter gradients (CPU, fp32, dropout off, ragged attention masks) -- this pins the restated BertLayer arithmetic
    that csrc/bertfuse.hip and csrc/attn_small.hip are tested against."""
    import pytest
    import torch
    transformers = pytest.importorskip('transformers')
    from creamfl_amd.networks import backbones as bb
    kw = dict(vocab_size=3000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
              max_position_embeddings=64)
    torch.manual_seed(0)
    hf = transformers.BertModel(transformers.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw),
                                add_pooling_layer=False).train()
    mine = bb.BertModel(bb.BertConfig(hidden_dropout_prob=0.0, **kw)).train()
    res = mine.load_state_dict(hf.state_dict(), strict=True)           # same parameter names
    assert not res.missing_keys and not res.unexpected_keys
    ids = torch.randint(1, 3000, (3, 11))
    mask = torch.arange(11)[None] < torch.tensor([11, 6, 9])[:, None]
    w = torch.randn(3, 128)
    a = hf(input_ids=ids, attention_mask=mask.long()).last_hidden_state
    b = mine(ids, attention_mask=mask)['last_hidden_state']
    torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
    (a[:, 0] * w).sum().backward()
    (b[:, 0] * w).sum().backward()
    ga = dict(hf.named_parameters())
    for n, p in mine.named_parameters():
        if n.endswith('key.bias'):
            continue                                                     # analytically zero
        g_ref = ga[n].grad
        assert (p.grad is None) == (g_ref is None), n
        if g_ref is not None:
            torch.testing.assert_close(p.grad, g_ref, rtol=1e-4, atol=1e-5 * (float(g_ref.abs().max()) + 1e-3), msg=n)
    c = mine(ids, attention_mask=mask, cls_only=True)['last_hidden_state'][:, 0]
    torch.testing.assert_close(c, a[:, 0].detach(), rtol=1e-5, atol=1e-5)


def test_main_py_flag_surface_drops_into_mmfl():
    """BASELINE north star: "keeping the src/algorithms client/server API surface so it drops into src/main.py unchanged".
    Build the Namespace main.py builds (all 41 flags at their defaults), construct MMFL exactly as main.py:118 does, set the
    two attributes main.py adds afterwards (:120-121) and check every name main.py touches (:118-134)."""
    from conftest import reference_main_namespace
    from creamfl_amd.algorithms.MMFL import MMFL
    args, spec = reference_main_namespace()
    assert len(spec['flags']) == 41 and args.pub_data_num == 50000 and args.feature_dim == 256 and args.agg_method == 'con_w'
    assert args.kd_weight == 0.3 and args.interintra_weight == 0.5 and args.client_num_per_round == 10
    algo = MMFL(args, None)                                   # main.py:118 (wandb object replaced by None)
    args.save_dirs = {'logs': '/tmp/creamfl_logs'}            # main.py:120-121
    args.log_dir = args.save_dirs['logs']
    for name in spec['algo_surface']:
        assert hasattr(algo, name), name
    algo.logger.log('ok')                                     # main.py:129
    # the config the constructor derives from the flags (MMFL.py:70-88): server ResNet-101 + BERT at feature_dim
    assert algo.config.model.embed_dim == 256 and algo.config.model.cnn_type == 'resnet101' and not algo.config.model.not_bert
    assert callable(algo.create_model) and callable(algo.load_dataset) and callable(algo.train)


def test_grad_buckets_views_layout_and_unused_parameters():
    """dist.GradBuckets on one process (no communication): after finish() every gradient is a view into its bucket with the
    parameter's own strides (channels_last convolution weights included), values equal plain autograd's, a parameter that
    received no gradient keeps `grad None` (optimizers must skip it as the single-process run does; its slot travels as
    zeros and consume() names it), and a gradient reported through `notify` (the deferred weight-gradient path) is bucketed
    like the hooked ones."""
    import torch.nn as nn
    from creamfl_amd.dist import GradBuckets
    torch.manual_seed(0)
    conv = nn.Conv2d(3, 8, 3).to(memory_format=torch.channels_last)
    lin = nn.Linear(8, 4)
    unused = nn.Parameter(torch.randn(5))
    deferred = nn.Parameter(torch.randn(6, 2))
    params = list(conv.parameters()) + list(lin.parameters()) + [unused, deferred]
    gb = GradBuckets(params, bucket_cap_mb=0.0005)
    assert len(gb.buckets) > 1
    x = torch.randn(2, 3, 6, 6)
    y = lin(conv(x).mean(dim=(2, 3))).sum()
    want = torch.autograd.grad(y, [conv.weight, conv.bias, lin.weight, lin.bias], retain_graph=True)
    gb.prepare()
    y.backward()
    deferred.grad = torch.full((6, 2), 3.0)               # what a deferred weight-gradient task does ...
    gb.notify(deferred)                                   # ... and how it reports
    gb.finish()
    for p, w in zip([conv.weight, conv.bias, lin.weight, lin.bias], want):
        assert torch.allclose(p.grad, w) and p.grad.stride() == p.stride()
    assert conv.weight.grad.is_contiguous(memory_format=torch.channels_last)
    assert unused.grad is None and torch.equal(gb.grad_views()[unused], torch.zeros(5))
    assert torch.equal(deferred.grad, torch.full((6, 2), 3.0))
    assert [id(p) for p in gb.consume()] == [id(unused)]
    for plist, views in zip(gb.buckets, gb.views):
        for p, v in zip(plist, views):
            assert p is unused or p.grad.data_ptr() == v.data_ptr()


def test_vit_patch_embedding_gemm_equals_the_convolution():
    """ViTTrunk.patch_embed (one GEMM over non-overlapping patches) == conv2d(kernel = stride = 16) on the same parameters:
    values and the gradients of the image, the weight and the bias; NCHW and channels_last inputs / weights."""
    from creamfl_amd.networks.backbones import ViTTrunk
    torch.manual_seed(0)
    m = ViTTrunk(dim=48, depth=1, heads=3, mlp_dim=96, patch=16, img=64).double()
    x = torch.randn(3, 3, 64, 64, dtype=torch.float64)
    for cl in (False, True):
        xi = (x.contiguous(memory_format=torch.channels_last) if cl else x).clone().requires_grad_(True)
        if cl:
            m.conv_proj.to(memory_format=torch.channels_last)
        m.zero_grad(set_to_none=True)
        t, h, w = m.patch_embed(xi)
        ref = m.conv_proj(xi).flatten(2).transpose(1, 2)
        assert (h, w) == (4, 4) and t.shape == ref.shape == (3, 16, 48)
        np.testing.assert_allclose(t.detach().numpy(), ref.detach().numpy(), rtol=1e-12, atol=1e-12)
        g = torch.randn_like(ref)
        gx, gw, gb = torch.autograd.grad((t * g).sum(), [xi, m.conv_proj.weight, m.conv_proj.bias])
        rx, rw, rb = torch.autograd.grad((ref * g).sum(), [xi, m.conv_proj.weight, m.conv_proj.bias])
        for a, b in ((gx, rx), (gw, rw), (gb, rb)):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-10, atol=1e-12)


def test_build_flags_register_on_the_reference_parser_and_default_to_reference_semantics():
    """ the build-defined flags have a parser (creamfl_amd/flags.py) and default to the reference's semantics:
    server phases replicated (full-batch BatchNorm statistics), fp32 wire; the reference's own Namespace (no such attributes)
    reads the same defaults."""
    import argparse
    from conftest import reference_main_namespace
    from creamfl_amd import flags
    ns, spec = reference_main_namespace()
    for name in flags.BUILD_FLAGS:
        assert not hasattr(ns, name) or name in {f['dest'] for f in spec['flags']}
    assert flags.get(ns, 'server_dp') == 0 and flags.get(ns, 'rep_wire') == 'fp32' and flags.get(ns, 'bucket_mb') == 32
    parser = argparse.ArgumentParser()
    parser.add_argument('--feature_dim', type=int, default=256)
    flags.add_build_flags(parser)
    flags.add_build_flags(parser)                              # idempotent
    a = parser.parse_args([])
    assert a.server_dp == 0 and a.rep_wire == 'fp32' and a.bucket_mb == 32 and a.quiet is False
    a = parser.parse_args(['--server_dp', '1', '--rep_wire', 'bf16', '--bucket_mb', '64'])
    assert flags.get(a, 'server_dp') == 1 and flags.get(a, 'rep_wire') == 'bf16' and flags.get(a, 'bucket_mb') == 64
    with pytest.raises(SystemExit):
        parser.parse_args(['--rep_wire', 'fp8'])


def test_device_prefetcher_passes_cpu_loaders_through_and_keeps_the_contract():
    """utils/prefetch.py on a CPU device is the loader itself (no thread, no copies); len() and .dataset are forwarded."""
    from creamfl_amd.utils.prefetch import DevicePrefetcher
    loader = SyntheticCocoLoader(10, 4, seed=3, img=8)
    pf = DevicePrefetcher(loader, 'cpu')
    assert len(pf) == len(loader) == 3 and pf.dataset is loader.dataset
    for a, b in zip(pf, loader):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[6] == b[6]


def test_caption_padding_of_the_captured_text_step_and_the_cpu_text_path():
    """Host logic around gru.hip: the one caption width a text client's captured step pads to; padding with the pad token;
    EncoderText's sentence states on the CPU (no HIP): the library formulation = the reference's lines, unchanged by extra
    padding (the packed recurrence never reaches padded words)."""
    import torch
    from creamfl_amd import ops
    from creamfl_amd.algorithms.ClientTrainer import caption_graph_width, pad_captions
    assert caption_graph_width(24) == 32 and caption_graph_width(21) == 32 and caption_graph_width(25) == 40
    assert caption_graph_width(3) == 32 and caption_graph_width(32) == 40 and caption_graph_width(57) % 8 == 0
    cap = torch.arange(12).view(2, 6) + 1
    p = pad_captions(cap, 9)
    assert p.shape == (2, 9) and torch.equal(p[:, :6], cap) and int(p[:, 6:].abs().sum()) == 0
    assert pad_captions(cap, 6) is cap and pad_captions(cap, 4) is cap
    # the CPU path never claims the HIP recurrence
    rnn = torch.nn.GRU(8, 128, bidirectional=True, batch_first=True)
    assert not ops.gru_last_supported(rnn) and not ops.gru_last_supported(rnn, torch.zeros(2, 3, 8))
    assert not ops.gru_last_supported(None)
    emb = torch.nn.Embedding(20, 8)
    tok = torch.randint(0, 20, (3, 5))
    assert torch.equal(ops.embedding_lookup(emb, tok), emb(tok))           # CPU: the module itself
    from creamfl_amd.networks.language_model import EncoderText
    torch.manual_seed(0)
    m = EncoderText(embed_dim=64, vocab_size=50)
    tokens = torch.randint(1, 50, (4, 7))
    lengths = torch.tensor([7, 5, 2, 1])
    tokens[torch.arange(7)[None] >= lengths[:, None]] = 0
    a, words = m.sentence_states(tokens, lengths)                          # (the PIE head behind it has no CPU path, by design)
    b, words_b = m.sentence_states(pad_captions(tokens, 12), lengths)
    assert a.shape == (4, 64) and words_b.shape == (4, 12, 300) and float((a - b).abs().max()) <= 1e-6


def test_adamp_capture_handles_count_replays_and_go_stale_on_the_host():
    """Host bookkeeping of AdamP steps replayed from a HIP graph (no kernel runs here): replays are folded into state[p]['step']
    lazily (state_dict / the next step), a handle goes stale when another captured step that skips one of its parameters is
    replayed, when a state dict is loaded, and capture_begin() refuses without prepare_capture()."""
    from creamfl_amd._lib import CreamflHipError
    from creamfl_amd.algorithms.optimizers import AdamP
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    opt = AdamP(ps, lr=1e-3)
    with pytest.raises(CreamflHipError):
        opt.capture_begin()
    opt.prepare_capture()
    assert int(opt._gstep_dev) == 0
    for p in ps:
        opt.state[p]['step'] = 5

    def captured(params):
        h = opt.capture_begin()
        h.params.extend(params)              # (what step() records while capturing)
        return opt.capture_end(h)
    train, kd = captured(ps), captured(ps[:2])          # the contrastive step updates all three, the KD step skips the last
    assert train.valid() and kd.valid()
    train.replayed(); train.replayed()
    assert kd.valid()                        # a superset was stepped: the KD step's offsets still hold
    assert [opt.state[p]['step'] for p in ps] == [5, 5, 5] and opt._gstep_host == 2
    steps = [st['step'] for st in opt.state_dict()['state'].values()]
    assert steps == [7, 7, 7] and train.pending == 0
    kd.replayed()
    assert not train.valid() and kd.valid()  # the third parameter fell behind the counter
    assert [st['step'] for st in opt.state_dict()['state'].values()] == [8, 8, 7]
    opt.load_state_dict(opt.state_dict())
    assert not kd.valid()


def test_adamp_failed_capture_rolls_back_on_the_host():
    """capture_end(handle, ok=False): the steps recorded inside a capture that failed are taken back from state[p]['step'] and from the
    host's running count (no kernel runs here: what step() records while capturing is written by hand)."""
    from creamfl_amd.algorithms.optimizers import AdamP
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(2)]
    opt = AdamP(ps, lr=1e-3)
    opt.prepare_capture()
    for p in ps:
        opt.state[p]['step'] = 4
    opt._gstep_host = 4
    h = opt.capture_begin()
    for p in ps:                                 # step() inside the capture: counts, remembers the parameters
        opt.state[p]['step'] += 1
    opt._gstep_host += 1
    h.nsteps += 1
    h.params.extend(ps)
    opt.capture_end(h, ok=False)
    assert [opt.state[p]['step'] for p in ps] == [4, 4] and opt._gstep_host == 4
    assert not h.valid() and h not in opt._live and opt._capture is None
    h2 = opt.capture_end(opt.capture_begin())   # a later capture is unaffected
    assert h2.valid()


def test_live_grad_node_probe_on_the_cpu():
    """graphs.live_grad_nodes: a referenced loss (with or without its buffers) keeps the parameters' AccumulateGrad nodes alive, a
    detached one does not -- the precondition GraphedStep checks before a capture."""
    from creamfl_amd.graphs import GraphedStep, live_grad_nodes
    lin = torch.nn.Linear(4, 4)
    assert live_grad_nodes(lin.parameters()) == 0
    loss = lin(torch.randn(2, 4)).sum()
    assert live_grad_nodes(lin.parameters()) == 2
    loss.backward()
    assert live_grad_nodes(lin.parameters()) == 2          # the buffers are gone, the graph's nodes are not
    del loss
    assert live_grad_nodes(lin.parameters()) == 0
    kept = lin(torch.randn(2, 4)).sum().detach()
    assert live_grad_nodes(lin.parameters()) == 0 and kept.grad_fn is None
    frozen = torch.nn.Linear(2, 2).requires_grad_(False)
    assert live_grad_nodes(frozen.parameters()) == 0
    gs = GraphedStep(lambda t: t, optimizer=torch.optim.SGD(lin.parameters(), lr=0.1), enabled=False)
    assert len(gs.guard_params) == 2


def test_server_graph_host_logic_on_the_cpu():
    """--server_graph host logic without a GPU: the engine refuses graphs on the CPU / with data parallel on / with a tokenizer that
    wants the caption strings; captions are padded to the graph's width (multiple of 8, fixed by the first batch) and lengths
    become int64; GraphedStep keys re-make the step when the learning rate changes and keep statistics of the dropped one."""
    from types import SimpleNamespace
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    eng = TrainerEngine(device='cpu')
    eng.server_graph = True
    assert not eng.graph_capable()                                     # CPU: never
    assert TrainerEngine.graph_caption_width(17) == 24 and TrainerEngine.graph_caption_width(24) == 24
    gs = SimpleNamespace(caption_width=None)
    im, cap, ln = eng.graph_inputs(gs, torch.zeros(2, 3, 4, 4), torch.ones(2, 19, dtype=torch.int64), torch.tensor([19, 7], dtype=torch.int32))
    assert gs.caption_width == 24 and cap.shape == (2, 24) and int(cap[:, 19:].abs().sum()) == 0 and ln.dtype == torch.int64
    _, cap2, _ = eng.graph_inputs(gs, torch.zeros(2, 3, 4, 4), torch.ones(2, 30, dtype=torch.int64), ln)
    assert cap2.shape == (2, 30)                                       # wider than the graph: handed through (that call runs eagerly)
    eng.drop_graph('train')                                            # nothing to drop: no statistics appear
    assert eng.graph_stats == {}
    made = []

    class FakeStep:
        def __init__(self, fn, **kw):
            made.append(kw)
            self.calls = self.replays = 0
            self.failed = None
    import creamfl_amd.graphs as graphs
    real, graphs.GraphedStep = graphs.GraphedStep, FakeStep
    try:
        eng.optimizer = object()
        a = eng.graphed_step('kd', ('lr', 1e-3), lambda: None)
        assert eng.graphed_step('kd', ('lr', 1e-3), lambda: None) is a and len(made) == 1
        a.calls, a.replays = 9, 5
        b = eng.graphed_step('kd', ('lr', 5e-4), lambda: None)         # another round's learning rate: a new step, the old one counted
        assert b is not a and eng.graph_stats['kd'] == {'calls': 9, 'replays': 5, 'failed': None}
        assert made[0]['other_threads'] is True and made[0]['optimizer'] is eng.optimizer
    finally:
        graphs.GraphedStep = real
