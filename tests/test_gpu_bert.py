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
