"""GPU: the fused BERT-tower glue (csrc/bertfuse.hip) against the eager composition it replaces
(transformers' BertSelfOutput / BertOutput / BertIntermediate, restated with torch ops in fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _ref_daln(g, bias, res, gamma, beta, keep, p, eps):
    """fp32 restatement on bf16-valued inputs; mirrors the kernel's single bf16 rounding of the pre-LN sum."""
    v = (g.float() + bias.float()) * keep.float() / (1.0 - p)
    s = (v + res.float()).to(torch.bfloat16).float()
    return F.layer_norm(s, (g.shape[-1],), gamma, beta, eps)


@pytest.mark.parametrize('t,h,p,bias_bf16', [(7, 256, 0.0, False), (64, 768, 0.1, True), (6144, 768, 0.1, True), (33, 1024, 0.25, False),
                                             (5, 2048, 0.0, True), (9, 260, 0.5, False)])
def test_daln_matches_eager(dev, t, h, p, bias_bf16):
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(t + h)
    g = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev).requires_grad_(True)
    res = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev).requires_grad_(True)
    bias = (torch.randn(h, generator=gen) * 0.5).to(torch.bfloat16 if bias_bf16 else torch.float32).to(dev).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(h, generator=gen)).to(dev).requires_grad_(True)
    beta = (0.3 * torch.randn(h, generator=gen)).to(dev).requires_grad_(True)
    seed = 1234 + t
    p_eff = round(p * 65536) / 65536                       # the kernel's 16-bit threshold
    za, zr = ops.bert_dropout_add_layernorm(g, bias, res, gamma, beta, p, 1e-12, seed=seed)
    assert za.data_ptr() == zr.data_ptr()
    wa = torch.randn(t, h, generator=gen).to(dev)
    wr = torch.randn(t, h, generator=gen).to(dev)
    ((za.float() * wa).sum() + (zr.float() * wr).sum()).backward()
    keep = ops.dropout_keep_mask(seed, p, (t, h), dev)
    if p > 0:
        assert abs(float(keep.float().mean()) - (1 - p_eff)) < 4 * (p_eff * (1 - p_eff) / (t * h)) ** 0.5 + 1e-3
    else:
        assert bool(keep.all())
    g2, r2, b2, ga2, be2 = [x.detach().clone().float().requires_grad_(True) for x in (g, res, bias, gamma, beta)]
    ref = _ref_daln(g2, b2, r2, ga2, be2, keep, p_eff, 1e-12)
    # the incoming gradients are bf16 in the product path (autograd casts the fp32 cotangents of .float())
    (ref * (wa.to(torch.bfloat16).float() + wr.to(torch.bfloat16).float())).sum().backward()
    np.testing.assert_allclose(za.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    scale = float(g2.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(g.grad.float().cpu().numpy(), g2.grad.cpu().numpy(), rtol=2e-2, atol=1.5e-2 * scale)
    np.testing.assert_allclose(res.grad.float().cpu().numpy(), r2.grad.cpu().numpy(), rtol=2e-2, atol=1.5e-2 * scale)
    for got, want, name in ((gamma.grad, ga2.grad, 'dgamma'), (beta.grad, be2.grad, 'dbeta'), (bias.grad.float(), b2.grad, 'dbias')):
        sc = float(want.abs().max()) + 1e-6
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-2, atol=1.5e-2 * sc, err_msg=name)
    if p > 0:                                              # dropped positions get exactly zero gradient
        assert bool((g.grad[~keep] == 0).all())


def test_daln_single_gradient_branch(dev):
    """only one of the two aliases is used downstream (the last layer): the other gradient is None"""
    from creamfl_amd import ops
    g = torch.randn(16, 768, device=dev).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(16, 768, device=dev).to(torch.bfloat16)
    gamma = torch.ones(768, device=dev, requires_grad=True)
    beta = torch.zeros(768, device=dev, requires_grad=True)
    za, zr = ops.bert_dropout_add_layernorm(g, None, res, gamma, beta, 0.0)
    zr.float().pow(2).sum().backward()
    g2 = g.detach().float().requires_grad_(True)
    F.layer_norm((g2 + res.float()).to(torch.bfloat16).float() + 0 * g2, (768,), gamma.detach(), beta.detach(), 1e-12).to(torch.bfloat16).float().pow(2).sum().backward()
    assert g.grad is not None and torch.isfinite(g.grad.float()).all()


@pytest.mark.parametrize('t,i,bias_bf16', [(3, 1024, False), (6144, 3072, True), (100, 4096, True), (17, 64, False)])
def test_bias_gelu_matches_eager(dev, t, i, bias_bf16):
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(t * 3 + i)
    g = (torch.randn(t, i, generator=gen) * 2).to(torch.bfloat16).to(dev).requires_grad_(True)
    bias = (torch.randn(i, generator=gen) * 0.5).to(torch.bfloat16 if bias_bf16 else torch.float32).to(dev).requires_grad_(True)
    w = torch.randn(t, i, generator=gen).to(torch.bfloat16).to(dev)
    h = ops.bert_bias_gelu(g, bias)
    h.backward(w)
    g2, b2 = g.detach().float().requires_grad_(True), bias.detach().float().requires_grad_(True)
    ref = F.gelu(g2 + b2)
    ref.backward(w.float())
    np.testing.assert_allclose(h.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(g.grad.float().cpu().numpy(), g2.grad.cpu().numpy(), rtol=1e-2, atol=2e-2)
    sc = float(b2.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(bias.grad.float().cpu().numpy(), b2.grad.cpu().numpy(), rtol=1e-2, atol=1e-2 * sc)


def test_bert_fused_tower_matches_library_path(dev):
    """BertModel under bf16 autocast: fused glue vs the plain torch modules (dropout off), forward and gradients."""
    from creamfl_amd.networks import backbones
    torch.manual_seed(0)
    m = backbones.BertModel('bert-mini').to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    ids = torch.randint(1, 1000, (8, 12), device=dev)
    mask = torch.arange(12, device=dev)[None] < torch.tensor([12, 5, 7, 9, 12, 3, 8, 11], device=dev)[:, None]
    wout = torch.randn(8, 256, device=dev)            # (|LN(x)|^2 is constant at gamma=1, beta=0: use a linear readout)

    def run(fused):
        orig = backbones._bert_fusable
        if not fused:
            backbones._bert_fusable = lambda *a, **k: False
        try:
            m.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = m(ids, attention_mask=mask, cls_only=True)['last_hidden_state'][:, 0].float()
            (out * wout).sum().backward()
            return out.detach(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            backbones._bert_fusable = orig
    o1, g1 = run(True)
    o0, g0 = run(False)
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=5e-2, atol=5e-2)
    assert set(g1) == set(g0)
    for n in g0:
        if n.endswith('key.bias'):
            continue                                     # analytically zero (softmax is shift-invariant): pure rounding noise
        sc = float(g0[n].abs().max()) + 1e-6
        err = float((g1[n] - g0[n]).abs().max())
        assert err <= 0.08 * sc + 1e-4, (n, err, sc)


@pytest.mark.parametrize('b,l,heads,masked', [(2, 24, 12, True), (256, 24, 12, True), (3, 32, 4, False), (5, 7, 2, True), (1, 1, 1, False)])
def test_attn_small_matches_sdpa(dev, b, l, heads, masked):
    """csrc/attn_small.hip against torch's scaled_dot_product_attention in fp32 on the same bf16 inputs
    (asymmetric random Q/K/V, ragged key-padding masks): forward and all three gradients."""
    from creamfl_amd import ops
    H = heads * 64
    gen = torch.Generator().manual_seed(b * 100 + l)
    qkv = (torch.randn(b, l, 3 * H, generator=gen) * 1.5).to(torch.bfloat16).to(dev).requires_grad_(True)
    lens = torch.randint(1, l + 1, (b,), generator=gen)
    lens[0] = l
    mask = (torch.arange(l)[None] < lens[:, None]).to(dev) if masked else None
    w = torch.randn(b, l, H, generator=gen).to(torch.bfloat16).to(dev)
    o = ops.bert_attention(qkv, mask, heads)
    o.backward(w)
    ref_in = qkv.detach().float().requires_grad_(True)
    q, k, v = ref_in.split(H, dim=-1)
    sp = lambda t: t.view(b, l, heads, 64).transpose(1, 2)
    am = mask[:, None, None, :] if masked else None
    ro = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=am).transpose(1, 2).reshape(b, l, H)
    ro.backward(w.float())
    np.testing.assert_allclose(o.detach().float().cpu().numpy(), ro.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    sc = float(ref_in.grad.abs().max()) + 1e-6
    np.testing.assert_allclose(qkv.grad.float().cpu().numpy(), ref_in.grad.cpu().numpy(), rtol=3e-2, atol=2e-2 * sc)


@pytest.mark.parametrize('b,l,heads', [(2, 24, 12), (256, 24, 12), (3, 32, 4), (5, 7, 2), (1, 1, 1)])
def test_attn_small_varlen_equals_the_padded_kernel(dev, b, l, heads):
    """The packed form of csrc/attn_small.hip (sequence b = rows cu[b] .. cu[b + 1] - 1 of [T, 3H]) against the padded kernel with its
    key-padding mask on the same tokens: per sequence the arithmetic is the same, so outputs and all three gradients agree BIT FOR
    BIT on every real token (the padded frame's masked rows have no counterpart)."""
    from creamfl_amd import ops
    from creamfl_amd.networks.backbones import BertModel
    H = heads * 64
    gen = torch.Generator().manual_seed(b * 100 + l)
    lens = torch.randint(1, l + 1, (b,), generator=gen)
    lens[0] = l
    plan = BertModel.pack_plan(lens.tolist(), l, dev)
    qkv_pad = (torch.randn(b, l, 3 * H, generator=gen) * 1.5).to(torch.bfloat16).to(dev)
    w_pad = torch.randn(b, l, H, generator=gen).to(torch.bfloat16).to(dev)
    mask = (torch.arange(l)[None] < lens[:, None]).to(dev)
    qp = qkv_pad.clone().requires_grad_(True)
    o_pad = ops.bert_attention(qp, mask, heads)
    # the gradient of a masked query row is multiplied by nothing downstream in the tower; here it must be zero to compare
    o_pad.backward(w_pad * mask[:, :, None])
    qv = qkv_pad.reshape(b * l, 3 * H).index_select(0, plan.tok_idx).clone().requires_grad_(True)
    assert qv.shape[0] == int(lens.sum()) == plan.T
    o_var = ops.bert_attention_varlen(qv, plan.cu, heads)
    o_var.backward(w_pad.reshape(b * l, H).index_select(0, plan.tok_idx))
    assert torch.equal(o_var, o_pad.reshape(b * l, H).index_select(0, plan.tok_idx))
    assert torch.equal(qv.grad, qp.grad.reshape(b * l, 3 * H).index_select(0, plan.tok_idx))


def test_bert_packed_tower_equals_the_padded_tower(dev):
    """BertModel on the batch's real tokens only (`pack`, round 6) against the reference's padded [B, L] frame with its attention
    mask (src/networks/models/pcme.py:43-57): the [CLS] states PCME reads and every parameter gradient agree to the bf16 rounding of
    GEMMs of another row count (the library picks other tiles for T rows than for B L rows); dropout off.  Also the full
    `last_hidden_state` of a call without `cls_only`, on the real tokens."""
    from creamfl_amd.networks import backbones
    torch.manual_seed(0)
    m = backbones.BertModel('bert-mini').to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    B, L = 16, 24
    gen = torch.Generator().manual_seed(5)
    lens = torch.randint(3, L + 1, (B,), generator=gen).sort(descending=True).values
    lens[0] = L
    ids = torch.randint(1, 1000, (B, L), generator=gen).to(dev)
    mask = (torch.arange(L)[None] < lens[:, None]).to(dev)
    ids = ids * mask                                     # 0-padded, as the loaders deliver
    plan = backbones.BertModel.pack_plan(lens.tolist(), L, dev)
    assert plan.T == int(lens.sum())
    wout = torch.randn(B, 256, device=dev)

    def run(pack, cls_only=True):
        m.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            h = m(ids, attention_mask=mask, cls_only=cls_only, pack=pack)['last_hidden_state']
        out = h[:, 0].float()
        (out * wout).sum().backward()
        return h.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
    h1, g1 = run(plan)
    h0, g0 = run(None)
    assert h1.shape == h0.shape == (B, 1, 256)
    np.testing.assert_allclose(h1.cpu().numpy(), h0.cpu().numpy(), rtol=3e-2, atol=3e-2)
    assert set(g1) == set(g0)
    for n in g0:
        if n.endswith('key.bias'):
            continue                                     # analytically zero: rounding noise
        a, b_ = g1[n].flatten(), g0[n].flatten()
        cos = float(torch.dot(a, b_) / (a.norm() * b_.norm() + 1e-30))
        sc = float(b_.abs().max()) + 1e-6
        assert cos >= 0.995 and float((a - b_).abs().max()) <= 0.06 * sc + 1e-4, (n, cos, float((a - b_).abs().max()), sc)
    hf1, _ = run(plan, cls_only=False)
    hf0, _ = run(None, cls_only=False)
    mk = mask[:, :, None].float()
    np.testing.assert_allclose((hf1 * mk).cpu().numpy(), (hf0 * mk).cpu().numpy(), rtol=3e-2, atol=3e-2)
    assert float((hf1 * (1 - mk)).abs().max()) == 0.0   # the packed run leaves the padded positions zero


def test_pcme_takes_the_packed_tower_only_with_host_lengths(dev):
    """PCME packs when the caption lengths are known on the host (a CPU tensor, or a device tensor carrying `_cfl_host_lens` as
    utils/synthetic.coco_batch and the prefetcher attach it) and keeps the padded frame for a bare device tensor (reading it would
    synchronise the step's issue thread); the two give the same caption embedding; the plan of a resident batch is built once."""
    from creamfl_amd.networks.models import pcme as pcme_mod
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    torch.manual_seed(0)
    cfg = default_config(embed_dim=64, cnn_type='resnet18', not_bert=False)
    cfg.model.bert_name = 'bert-mini'
    model = pcme_mod.PCME({'<pad>': 0}, cfg.model, False).to(dev).eval()
    b = coco_batch(8, dev, seed=3, img=64)
    lens = b[3]
    assert lens.is_cuda and lens._cfl_host_lens == tuple(lens.tolist())
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        packed = model._bert_inputs(b[1], None, lens)
        assert 'pack' in packed and packed['pack'].T == int(lens.sum())
        assert model._bert_inputs(b[1], None, lens)['pack'] is packed['pack']            # cached for the resident batch
        bare = lens.clone()                                                                # a device tensor without a host copy
        assert 'pack' not in model._bert_inputs(b[1], None, bare)
        e1 = model._text_tower(b[1], None, lens)['embedding'].float()
        e0 = model._text_tower(b[1], None, bare)['embedding'].float()
    assert float((e1 - e0).abs().max()) <= 3e-2
    assert 'pack' in model._bert_inputs(b[1], None, lens.cpu())                            # host tensor: no copy needed


@pytest.mark.parametrize('t,h,bias_bf16,direct', [(7, 256, False, True), (50176, 768, True, True), (197, 192, True, False), (33, 1024, False, True),
                                                  (9, 260, True, True)])
def test_preln_add_layernorm_matches_eager(dev, t, h, bias_bf16, direct):
    """Round 6, the pre-LN tail of a ViT sub-layer (ops.preln_add_layernorm): (s, z) = (g + bias + residual, LayerNorm(s)) against the
    fp32 composition on the same bf16-valued inputs; s takes a gradient of its own (the residual stream) unless `direct` is off."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(t + h)
    g = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev).requires_grad_(True)
    res = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev).requires_grad_(True)
    bias = (0.3 * torch.randn(h, generator=gen)).to(torch.bfloat16 if bias_bf16 else torch.float32).to(dev).requires_grad_(True)
    gamma = (1.0 + 0.2 * torch.randn(h, generator=gen)).to(dev).requires_grad_(True)
    beta = (0.2 * torch.randn(h, generator=gen)).to(dev).requires_grad_(True)
    wz = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev)
    ws = torch.randn(t, h, generator=gen).to(torch.bfloat16).to(dev)
    s, z = ops.preln_add_layernorm(g, bias, res, gamma, beta, 1e-6)
    loss = (z.float() * wz.float()).sum() + ((s.float() * ws.float()).sum() if direct else 0.0)
    loss.backward()
    got = [x.grad.float().clone() for x in (g, res, bias, gamma, beta)]
    r = [x.detach().float().requires_grad_(True) for x in (g, res, bias, gamma, beta)]
    s0 = (r[0] + r[2] + r[1])
    s0q = s0 + (s0.detach().to(torch.bfloat16).float() - s0.detach())          # the kernel's single bf16 rounding of the sum (straight through)
    z0 = F.layer_norm(s0q, (h,), r[3], r[4], 1e-6)
    ((z0 * wz.float()).sum() + ((s0q * ws.float()).sum() if direct else 0.0)).backward()
    np.testing.assert_allclose(s.detach().float().cpu().numpy(), s0q.detach().cpu().numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(z.detach().float().cpu().numpy(), z0.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    assert torch.equal(g.grad, res.grad)
    for name, a, b in zip(('g', 'res', 'bias', 'gamma', 'beta'), got, [x.grad for x in r]):
        sc = float(b.abs().max()) + 1e-6
        err = float((a - b).abs().max())
        assert err <= 2e-2 * sc, (name, err, sc)             # bf16 gradients / fp32 column sums of bf16-rounded rows


def test_vit_fused_chain_matches_the_aten_blocks(dev, monkeypatch):
    """Round 6 (configs[4]): the ViT trunk on the fused pre-LN glue (each block's tail forms the next block's ln_1; bias + GELU fused)
    against the same trunk on aten LayerNorm / GELU / adds under bf16 autocast: features and every parameter gradient."""
    from creamfl_amd import _lib
    from creamfl_amd.networks import backbones
    torch.manual_seed(1)
    vit = backbones.ViTTrunk(dim=192, depth=3, heads=3, mlp_dim=512, patch=16, img=64).to(dev).train()
    x = torch.randn(5, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    wout = torch.randn(5, 192, 4, 4, device=dev)

    def run(fused):
        monkeypatch.setattr(backbones, '_VIT_NO_FUSE', not fused)
        vit.zero_grad(set_to_none=True)
        _lib.prof_enable(True)
        _lib.prof_reset()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = vit(x).float()
        (out * wout).sum().backward()
        launches = {k: v[0] for k, v in _lib.prof_query().items()}
        _lib.prof_enable(False)
        return out.detach(), {n: p.grad.detach().float().clone() for n, p in vit.named_parameters()}, launches
    o1, g1, l1 = run(True)
    o0, g0, l0 = run(False)
    assert l1.get('cfl_bert_daln_kernel', 0) >= 2 * 2 * 3 and l0.get('cfl_bert_daln_kernel', 0) == 0, (l1, l0)
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=5e-2, atol=5e-2)
    assert set(g1) == set(g0)
    for n in g0:
        sc = float(g0[n].abs().max()) + 1e-6
        err = float((g1[n] - g0[n]).abs().max())
        cos = float(torch.nn.functional.cosine_similarity(g1[n].flatten(), g0[n].flatten(), dim=0))
        assert err <= 0.1 * sc + 1e-4 and cos >= 0.99, (n, err, sc, cos)
