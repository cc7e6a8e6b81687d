"""The oracle (oracle/*.py) against the golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.mark.parametrize('fname', golden_files('a1_'))
def test_a1_pair_loss_literal(fname):
    z = _load(fname)
    I = torch.from_numpy(z['I']).requires_grad_(True)
    T = torch.from_numpy(z['T']).requires_grad_(True)
    a = torch.tensor([float(z['a'])], requires_grad=True)
    b = torch.tensor([float(z['b'])], requires_grad=True)
    loss, ld = oracle.pair_loss_literal(I, T, a, b)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=1e-6)
    ref = dict(zip([str(k) for k in z['dict_keys']], z['dict_vals']))
    assert list(ld.keys()) == list(ref.keys())
    for k in ref:
        np.testing.assert_allclose(ld[k], ref[k], rtol=2e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(I.grad.numpy(), z['dI'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(T.grad.numpy(), z['dT'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.grad.numpy(), z['da'], rtol=1e-5)
    np.testing.assert_allclose(b.grad.numpy(), z['db'], rtol=1e-5)
    mp = oracle.match_prob(I.detach(), T.detach(), a.detach(), b.detach())
    np.testing.assert_allclose(mp.numpy(), z['match_prob'], rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize('fname', golden_files('a1_'))
def test_a1_closed_form_matches_reference(fname):
    """fp64 closed form vs. the reference's fp32 result: the north-star tolerance
    (loss rel 1e-4), grads to fp32 round-off."""
    z = _load(fname)
    I, T = torch.from_numpy(z['I']), torch.from_numpy(z['T'])
    a, b = float(z['a']), float(z['b'])
    cf = oracle.pair_loss_closed_form(I, T, a, b)
    # The reference evaluates NLL = logsumexp(s,-s) - m*s in fp32: when |s| ~ 15 the two
    # terms cancel, leaving up to ~ulp(15)/2 = 5e-7 of noise PER PAIR (probemb.py:82-86).
    # The fp64 closed form is the exact value; the reference agrees to rtol 1e-5 plus that
    # per-pair round-off budget (which only matters when the total loss is tiny).
    n = I.shape[0]
    noise = n * n * 2.5e-7
    np.testing.assert_allclose(cf['loss'].item(), float(z['loss']), rtol=1e-5, atol=2 * noise)
    ref = dict(zip([str(k) for k in z['dict_keys']], z['dict_vals']))
    np.testing.assert_allclose(cf['pos'].item(), ref['i2t_pos_loss'], rtol=1e-5, atol=n * 5e-7)
    np.testing.assert_allclose(cf['neg'].item(), ref['i2t_neg_loss'], rtol=1e-5, atol=noise)
    g = oracle.pair_loss_grads_closed_form(I, T, a, b)
    # Reference autograd evaluates d/ds[logsumexp(s,-s) + s] = (p0 - p1) + 1 in fp32, which
    # cancels to ~6e-8 absolute per negative pair; gradients therefore agree with the exact
    # closed form only to ~1e-3 of their scale ("grads rel 1e-3").
    # Budget per row: n pairs x ulp(1) x a (|I-T|/d <= 1).
    scale = np.abs(z['dI']).max()
    atol = max(1e-3 * scale, n * a * 1.2e-7)
    np.testing.assert_allclose(g['dI'].numpy(), z['dI'], rtol=1e-4, atol=atol)
    np.testing.assert_allclose(g['dT'].numpy(), z['dT'], rtol=1e-4, atol=atol)
    np.testing.assert_allclose(g['da'].item(), float(z['da'][0]), rtol=1e-3, atol=n * n * 2.4e-7)
    np.testing.assert_allclose(g['db'].item(), float(z['db'][0]), rtol=1e-3, atol=n * n * 2.4e-7)


@pytest.mark.parametrize('fname', golden_files('a2_'))
def test_a2_pie_head(fname):
    z = _load(fname)
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = torch.from_numpy(z['out']).requires_grad_(True)
    mask = torch.from_numpy(z['mask']) if z['mask'].size else None
    names = ['attention__w_1__weight', 'attention__w_2__weight', 'fc__weight', 'fc__bias',
             'layer_norm__weight', 'layer_norm__bias']
    w1, w2, fcw, fcb, lnw, lnb = [torch.from_numpy(z['p_' + n]).requires_grad_(True) for n in names]
    o, attn, res = oracle.pie_head(out, x, w1, w2, fcw, fcb, lnw, lnb, pad_mask=mask)
    y = oracle.l2_normalize(o)
    (y * torch.from_numpy(z['gy'])).sum().backward()
    np.testing.assert_allclose(o.detach().numpy(), z['o'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(attn.detach().numpy(), z['attn'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res.detach().numpy(), z['res'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.detach().numpy(), z['y'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), z['dx'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(out.grad.numpy(), z['dout'], rtol=1e-4, atol=1e-7)
    for n, p in zip(names, [w1, w2, fcw, fcb, lnw, lnb]):
        np.testing.assert_allclose(p.grad.numpy(), z['g_' + n], rtol=1e-4, atol=1e-6, err_msg=n)


@pytest.mark.parametrize('fname', golden_files('a34_'))
def test_a34_client_contrast(fname):
    z = _load(fname)
    f = torch.from_numpy(z['f']).requires_grad_(True)
    args = (torch.from_numpy(z['g_same']), torch.from_numpy(z['g_other']),
            tuple(int(v) for v in z['d_idx']), torch.from_numpy(z['f_old']))
    loss, li, lm = oracle.client_contrast_loss(f, *args, interintra_weight=float(z['weight']),
                                               loss_scale=bool(z['loss_scale']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=1e-6)
    np.testing.assert_allclose(li.item(), float(z['loss_inter']), rtol=1e-6)
    np.testing.assert_allclose(lm.item(), float(z['loss_moon']), rtol=1e-6)
    np.testing.assert_allclose(f.grad.numpy(), z['df'], rtol=1e-5, atol=1e-8)
    f2 = torch.from_numpy(z['f']).requires_grad_(True)
    oracle.client_contrast_loss(f2, *args, use_intra=False)[0].backward()
    np.testing.assert_allclose(f2.grad.numpy(), z['df_inter_only'], rtol=1e-5, atol=1e-8)
    f3 = torch.from_numpy(z['f']).requires_grad_(True)
    oracle.client_contrast_loss(f3, *args, use_inter=False)[0].backward()
    np.testing.assert_allclose(f3.grad.numpy(), z['df_intra_only'], rtol=1e-5, atol=1e-8)
    # closed forms (fp64) agree with the reference's fp32 numbers
    cf = oracle.client_contrast_grads_closed_form(torch.from_numpy(z['f']), *args)
    np.testing.assert_allclose(cf['loss_inter'].item(), float(z['loss_inter']), rtol=1e-5)
    np.testing.assert_allclose(cf['loss_moon'].item(), float(z['loss_moon']), rtol=1e-5)
    np.testing.assert_allclose(cf['d_inter'].numpy(), z['df_inter_only'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(cf['d_moon'].numpy(), z['df_intra_only'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('fname', golden_files('a34mm_'))
def test_a34_mm_client_contrast(fname):
    """oracle.mm_client_contrast_loss vs the literal statement sequences of MMClientTrainer.py:164-206 (both terms, with and
    without --loss_scale), :246-264 (intra only), :301-308 (inter only) evaluated with the reference's criterion object."""
    z = _load(fname)
    d_idx = tuple(int(v) for v in z['d_idx'])
    consts = (torch.from_numpy(z['g_img']), torch.from_numpy(z['g_txt']), d_idx, torch.from_numpy(z['old_img']),
              torch.from_numpy(z['old_txt']))

    def leaves():
        return torch.from_numpy(z['out_img']).requires_grad_(True), torch.from_numpy(z['out_txt']).requires_grad_(True)

    i, t = leaves()
    loss, li, lm = oracle.mm_client_contrast_loss(i, t, *consts, interintra_weight=float(z['weight']),
                                                  loss_scale=bool(z['loss_scale']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=1e-6)
    np.testing.assert_allclose(li.item(), float(z['loss_inter']), rtol=1e-6)
    np.testing.assert_allclose(lm.item(), float(z['loss_intra']), rtol=1e-6)
    np.testing.assert_allclose(i.grad.numpy(), z['d_img'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(t.grad.numpy(), z['d_txt'], rtol=1e-5, atol=1e-8)
    i, t = leaves()
    loss, li, lm = oracle.mm_client_contrast_loss(i, t, *consts, use_inter=False)
    assert li is None
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss_intra_only']), rtol=1e-6)
    np.testing.assert_allclose(i.grad.numpy(), z['d_img_intra_only'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(t.grad.numpy(), z['d_txt_intra_only'], rtol=1e-5, atol=1e-8)
    i, t = leaves()
    loss, li, lm = oracle.mm_client_contrast_loss(i, t, *consts[:3], use_intra=False)
    assert lm is None
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss_inter_only']), rtol=1e-6)
    np.testing.assert_allclose(i.grad.numpy(), z['d_img_inter_only'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(t.grad.numpy(), z['d_txt_inter_only'], rtol=1e-5, atol=1e-8)
    # the fp64 closed forms of the two modalities assemble the same numbers (what the GPU tests compare against)
    g_img, g_txt, _, o_img, o_txt = consts
    ci = oracle.client_contrast_grads_closed_form(torch.from_numpy(z['out_img']), g_img, g_txt, d_idx, o_img)
    ct = oracle.client_contrast_grads_closed_form(torch.from_numpy(z['out_txt']), g_txt, g_img, d_idx, o_txt)
    np.testing.assert_allclose((ci['loss_inter'] + ct['loss_inter']).item(), float(z['loss_inter']), rtol=1e-5)
    np.testing.assert_allclose(((ci['loss_moon'] + ct['loss_moon']) / 2).item(), float(z['loss_intra']), rtol=1e-5)
    np.testing.assert_allclose(ci['d_inter'].numpy(), z['d_img_inter_only'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ct['d_moon'].numpy() / 2, z['d_txt_intra_only'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('fname', golden_files('a5_'))
@pytest.mark.parametrize('literal', [True, False])
def test_a5_conw(fname, literal):
    z = _load(fname)
    vecs = [torch.from_numpy(v) for v in z['vecs']]
    agg, w, lp = oracle.conw_aggregate(vecs, torch.from_numpy(z['g_other']), literal=literal)
    tol = 1e-6 if literal else 1e-5
    np.testing.assert_allclose(lp.numpy(), z['logprob'], rtol=tol, atol=tol)
    np.testing.assert_allclose(w.numpy(), z['weights'], rtol=10 * tol, atol=tol)
    np.testing.assert_allclose(agg.numpy(), z['agg'], rtol=10 * tol, atol=tol)


@pytest.mark.parametrize('fname', golden_files('a6_'))
def test_a6_recall(fname):
    z = _load(fname)
    keys = [str(k) for k in z['keys']]
    for (q, g, ql, gl, want) in [(z['img'], z['cap'], z['img_cls'], z['cap_cls'], z['i2t']),
                                 (z['cap'], z['img'], z['cap_cls'], z['img_cls'], z['t2i'])]:
        lit = oracle.recall_ranks_literal(q, g, ql, gl, n_embeddings=7, batch_size=64)
        cnt = oracle.recall_ranks_count(q, g, ql, gl)
        assert np.array_equal(lit, cnt)
        sc = oracle.recall_scores(cnt)
        np.testing.assert_array_equal(np.array([sc[k] for k in keys]), want)


@pytest.mark.parametrize('fname', golden_files('f4_'))
def test_f4_supervised_glue(fname):
    """oracle/supervised.py against the reference's to_one_hot + criterion statement sequence."""
    from oracle import supervised
    z = _load(fname)
    fvec = torch.from_numpy(z['fvec']).requires_grad_(True)
    W = torch.from_numpy(z['class_weight']).requires_grad_(True)
    labels = torch.from_numpy(z['labels'])
    total, ce, center, p1, pk = supervised.supervised_glue(fvec, labels, W, float(z['margin']), int(z['topk']))
    total.backward()
    np.testing.assert_allclose(total.item(), float(z['total']), rtol=1e-6)
    np.testing.assert_allclose(ce.item(), float(z['ce']), rtol=1e-6)
    np.testing.assert_allclose(center.item(), float(z['center']), rtol=1e-6)
    assert float(p1) == float(z['prec1'][0]) and float(pk) == float(z['preck'][0])
    np.testing.assert_allclose(fvec.grad.numpy(), z['dfvec'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(W.grad.numpy(), z['dclass_weight'], rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------ A2c / tower glue / KD (round 2 fixtures)
import sys
sys.path.insert(0, GOLDEN)
from seeded import (seeded_state_dict, checksum, resnet_client_template, text_client_template,  # noqa: E402
                    tower_template)


def _gradkeys(z, prefix):
    return {k[len(prefix):].replace('__', '.'): z[k] for k in z.files if k.startswith(prefix)}


def _scale_close(got, want, rel, msg=''):
    want = np.asarray(want, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=rel, atol=rel * (np.abs(want).max() + 1e-30),
                               err_msg=msg)



def _leafify(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in sd.items()}


@pytest.mark.parametrize('fname', golden_files('a2c_img_'))
def test_a2c_resnet_client_both_phases(fname):
    """oracle.resnet_client_forward vs the reference's resnet_client.ResNet.forward (:175-201), both phases, incl. the
    BatchNorm running-statistics update and the `weight.data = relu(weight)` side effect."""
    z = _load(fname)
    d, train = int(z['embed_dim']), bool(z['train'])
    sd = seeded_state_dict(resnet_client_template(d), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)          # same weights as the generator's
    x = torch.from_numpy(z['x'])
    p = _leafify(sd)
    feat, new = oracle.resnet_client_forward(p, x, 'extract_conv_feature', train_mode=train)
    _scale_close(feat.detach().numpy(), z['feat'], 2e-5)
    (feat * torch.from_numpy(z['gy'])).sum().backward()
    for k, g in _gradkeys(z, 'feat__g_').items():
        _scale_close(p[k].grad.numpy(), g, 2e-4, k)
    if train:
        _scale_close(new['bn1.running_mean'].numpy(), z['bn1_running_mean_after_feat'], 1e-5)
        sd['bn1.running_mean'] = new['bn1.running_mean']                               # the reference's second forward starts here
    p = _leafify({**sd, **{k: v for k, v in new.items() if 'running' in k}})
    (x1, x2, w, w2), new2 = oracle.resnet_client_forward(p, x, 'none', is_train=True, train_mode=train)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (w, 'w'), (w2, 'w2')):
        _scale_close(got.detach().numpy(), z[key], 2e-5, key)
    np.testing.assert_array_equal(new2['class_fc_2.weight'].numpy(), z['class_fc_2_weight_after'])
    np.testing.assert_array_equal(new2['class_fc_22.weight'].numpy(), z['class_fc_22_weight_after'])
    assert (z['class_fc_2_weight_after'] >= 0).all() and (sd['class_fc_2.weight'].numpy() < 0).any()
    ((x1 * torch.from_numpy(z['g1'])).sum() + (x2 * torch.from_numpy(z['g2'])).sum() + 0.1 * (w ** 2).sum()).backward()
    for k, g in _gradkeys(z, 'cls__g_').items():
        _scale_close(p[k].grad.numpy(), g, 2e-4, k)


@pytest.mark.parametrize('fname', golden_files('a2c_txt_'))
def test_a2c_text_client_both_modes(fname):
    """oracle.text_client_forward vs the reference's language_model.EncoderText.forward (:93-130)."""
    z = _load(fname)
    d = int(z['embed_dim'])
    sd = seeded_state_dict(text_client_template(d, int(z['vocab'])), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)
    x, lengths = torch.from_numpy(z['x']), torch.from_numpy(z['lengths'])
    p = _leafify(sd)
    feat, _ = oracle.text_client_forward(p, x, lengths, is_train=False)
    _scale_close(feat.detach().numpy(), z['feat'], 2e-5)
    (feat * torch.from_numpy(z['gy'])).sum().backward()
    for k, g in _gradkeys(z, 'feat__g_').items():
        _scale_close(p[k].grad.numpy(), g, 2e-4, k)
    p = _leafify(sd)
    (x1, x2, w, w2), new = oracle.text_client_forward(p, x, lengths, is_train=True)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (w, 'w'), (w2, 'w2')):
        _scale_close(got.detach().numpy(), z[key], 2e-5, key)
    np.testing.assert_array_equal(new['class_fc.weight'].numpy(), z['class_fc_weight_after'])
    ((x1 * torch.from_numpy(z['g1'])).sum() + (x2 * torch.from_numpy(z['g2'])).sum() + 0.1 * (w ** 2).sum()).backward()
    for k, g in _gradkeys(z, 'cls__g_').items():
        _scale_close(p[k].grad.numpy(), g, 2e-4, k)


@pytest.mark.parametrize('fname', golden_files('tower_'))
def test_a2_pcme_tower_glue(fname):
    """oracle.pcme_towers_forward vs the reference's own PCME.forward / EncoderImage.forward / EncoderText.forward run on a
    synthetic trunk output: 10-key dict, fc(avgpool) + PIE + (head_proj) + l2norm ordering of both towers."""
    z = _load(fname)
    cd, d, mlp = int(z['cd']), int(z['embed_dim']), bool(z['mlp_local'])
    sd = seeded_state_dict(tower_template(cd, d, mlp), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)
    p = _leafify(sd)
    fmap = torch.from_numpy(z['fmap']).requires_grad_(True)
    out = oracle.pcme_towers_forward(p, fmap, torch.from_numpy(z['sentences']), torch.from_numpy(z['lengths']), mlp_local=mlp)
    assert list(out.keys()) == [str(k) for k in z['keys']]
    assert [k for k in out if out[k] is None] == [str(k) for k in z['none_keys']]
    _scale_close(out['image_features'].detach().numpy(), z['image_features'], 2e-5)
    _scale_close(out['caption_features'].detach().numpy(), z['caption_features'], 2e-5)
    ((out['image_features'] * torch.from_numpy(z['gi'])).sum() + (out['caption_features'] * torch.from_numpy(z['gc'])).sum()).backward()
    _scale_close(fmap.grad.numpy(), z['dfmap'], 2e-4)
    for k, g in _gradkeys(z, 'g_').items():
        _scale_close(p[k].grad.numpy(), g, 2e-4, k)


@pytest.mark.parametrize('fname', golden_files('kd_'))
def test_f1_kd_terms(fname):
    """oracle.kd_loss vs the literal MMFL.py:346-378 sequence (incl. the doubled image term)."""
    z = _load(fname)
    oi = torch.from_numpy(z['out_img']).requires_grad_(True)
    ot = torch.from_numpy(z['out_txt']).requires_grad_(True)
    dd = {int(b): a for a, b in enumerate(z['distill_index'])}
    d_idx = [dd[int(i)] for i in z['index']]
    assert d_idx == [int(v) for v in z['d_idx']]
    loss = oracle.kd_loss(oi, ot, torch.from_numpy(z['img_vec']), torch.from_numpy(z['txt_vec']), d_idx,
                          int(z['num_img_clients']), int(z['num_txt_clients']), int(z['num_mm_clients']), float(z['kd_weight']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z['loss']), rtol=1e-6)
    _scale_close((oi.grad if oi.grad is not None else torch.zeros_like(oi)).numpy(), z['d_out_img'], 1e-6)
    _scale_close((ot.grad if ot.grad is not None else torch.zeros_like(ot)).numpy(), z['d_out_txt'], 1e-6)


# ---- A2c: the text towers' recurrence, last valid step only (oracle/gru.py) -----------------------------------------------------
@pytest.mark.parametrize('hidden,B,T', [(16, 5, 7), (128, 9, 12)])
def test_gru_last_states_restatement_matches_the_reference_lines(hidden, B, T):
    """The numpy restatement of torch.nn.GRU (oracle/gru.py) against the reference's own three lines run through torch on the CPU
    (pack -> bi-GRU -> pad -> gather at lengths - 1, language_model.py:99-107): pins the oracle AND the property gru.hip is built
    on -- at the gathered position the backward direction has taken exactly one step from a zero state."""
    import torch
    from oracle import gru as ogru
    torch.manual_seed(3)
    rnn = torch.nn.GRU(20, hidden, bidirectional=True, batch_first=True)
    words = torch.randn(B, T, 20)
    lengths = torch.tensor(sorted(np.random.RandomState(1).randint(1, T + 1, size=B).tolist(), reverse=True))
    lengths[0] = T
    lengths[-1] = 1
    want = ogru.reference_formulation(rnn, words, lengths).detach().numpy()
    got = ogru.bigru_last_states(words.numpy(), lengths.numpy(), ogru.gru_params(rnn))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    # the gradient of the backward direction's hidden-side weight is identically zero (and a tensor, not None)
    ogru.reference_formulation(rnn, words, lengths).square().sum().backward()
    assert rnn.weight_hh_l0_reverse.grad is not None and float(rnn.weight_hh_l0_reverse.grad.abs().max()) == 0.0
    assert float(rnn.weight_hh_l0.grad.abs().max()) > 0.0
