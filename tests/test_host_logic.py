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
