"""GPU: the text towers' recurrence, last valid step only (csrc/gru.hip through ops.bigru_last_states), against the reference's
own formulation -- pack_padded_sequence -> nn.GRU(bidirectional) -> pad_packed_sequence -> gather(lengths - 1)
(language_model.py:99-107) -- evaluated by torch on the CPU in fp32, and against the fp64 restatement of oracle/gru.py.
Tolerances: values 2e-6 absolute (states are in (-1, 1); fp32 FMA chains in a different order than the library's GEMM), gradients
2e-5 of each tensor's largest reference entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PARAMS = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse',
          'bias_ih_l0_reverse', 'bias_hh_l0_reverse']


def _case(hidden, B, T, E, seed, full=False):
    g = torch.Generator().manual_seed(seed)
    rnn = torch.nn.GRU(E, hidden, bidirectional=True, batch_first=True)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / hidden ** 0.5 if p.dim() == 2 else 0.2))
    words = torch.randn(B, T, E, generator=g)
    lengths = torch.full((B,), T, dtype=torch.int64) if full else \
        torch.tensor(sorted(torch.randint(1, T + 1, (B,), generator=g).tolist(), reverse=True))
    if not full and B > 1:
        lengths[-1] = 1
    gy = torch.randn(B, 2 * hidden, generator=g)
    return rnn, words, lengths, gy


@pytest.mark.parametrize('hidden,B,T,E', [(128, 128, 31, 300), (128, 7, 12, 300), (64, 10, 9, 40), (32, 5, 12, 300), (128, 1, 5, 16),
                                         (128, 32, 57, 300),
                                         # widths beyond the register-resident kernels: W_hh streamed from L2 (d = 512 / 768 text towers)
                                         (256, 20, 14, 300), (384, 9, 11, 300), (16, 5, 7, 20), (100, 6, 9, 24), (512, 3, 6, 32),
                                         (256, 128, 24, 300)])
@pytest.mark.parametrize('lengths_on', ['host', 'device'])
def test_bigru_last_states_match_the_reference_lines(hidden, B, T, E, lengths_on):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    from oracle import gru as ogru
    dev = torch.device('cuda:0')
    rnn, words, lengths, gy = _case(hidden, B, T, E, seed=hidden + B + T)
    # reference lines on the CPU (torch's own GRU)
    wr = words.clone().requires_grad_(True)
    want = ogru.reference_formulation(rnn, wr, lengths)
    (want * gy).sum().backward()
    ref_grads = {k: getattr(rnn, k).grad.clone() for k in PARAMS}
    # HIP path
    rnn_d = torch.nn.GRU(E, hidden, bidirectional=True, batch_first=True)
    rnn_d.load_state_dict(rnn.state_dict())
    rnn_d = rnn_d.to(dev)
    wd = words.to(dev).requires_grad_(True)
    assert ops.gru_last_supported(rnn_d, wd)
    got = ops.bigru_last_states(rnn_d, wd, lengths.to(dev) if lengths_on == 'device' else lengths)
    (got * gy.to(dev)).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=2e-6)
    # and the fp64 restatement (small cases: pure-python loops)
    if B * T <= 400:
        o = ogru.bigru_last_states(words.numpy(), lengths.numpy(), ogru.gru_params(rnn))
        np.testing.assert_allclose(got.detach().cpu().numpy(), o, rtol=0, atol=2e-6)

    def close(a, b, name):
        scale = float(b.abs().max())
        err = float((a.cpu() - b).abs().max())
        assert err <= 2e-5 * max(scale, 1e-3), f'{name}: |diff| {err:.3e} at scale {scale:.3e}'
    close(wd.grad, wr.grad, 'words')
    for k in PARAMS:
        gk = getattr(rnn_d, k).grad
        assert gk is not None, k                       # weight_hh_l0_reverse: a zero TENSOR like the reference's, not None
        close(gk, ref_grads[k], k)
    assert float(rnn_d.weight_hh_l0_reverse.grad.abs().max()) == 0.0
    # padded positions receive no gradient, exactly
    mask = torch.arange(T).view(1, T) >= lengths.view(B, 1)
    assert float(wd.grad.cpu()[mask].abs().max() if mask.any() else 0.0) == 0.0


def test_bigru_last_states_forward_only_and_padding_independence():
    """No-grad call (representation extraction): same values, no saved state; values do not depend on what the padded positions
    hold nor on how wide the padding is (the captured client step pads every batch to one width)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    dev = torch.device('cuda:0')
    rnn, words, lengths, _ = _case(128, 20, 15, 300, seed=5)
    rnn = rnn.to(dev)
    wd = words.to(dev)
    with torch.no_grad():
        a = ops.bigru_last_states(rnn, wd, lengths)
        junk = wd.clone()
        mask = (torch.arange(15).view(1, 15) >= lengths.view(20, 1)).to(dev)
        junk[mask] = 1e30
        b = ops.bigru_last_states(rnn, junk, lengths.to(dev))
        wide = torch.cat([wd, torch.full((20, 9, 300), float('nan'), device=dev)], 1)
        c = ops.bigru_last_states(rnn, wide, lengths.to(dev))
    assert torch.equal(a, b) and torch.equal(a, c)
    g = ops.bigru_last_states(rnn, wd.clone().requires_grad_(True), lengths)
    assert torch.equal(a, g.detach())


def test_bigru_last_states_host_lengths_fail_like_pack_padded_sequence():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    dev = torch.device('cuda:0')
    rnn, words, lengths, _ = _case(128, 6, 8, 300, seed=9)
    rnn, wd = rnn.to(dev), words.to(dev)
    with pytest.raises(RuntimeError, match='sorted in decreasing order'):
        ops.bigru_last_states(rnn, wd, torch.tensor([3, 5, 2, 2, 1, 1]))
    with pytest.raises(RuntimeError, match='greater than 0'):
        ops.bigru_last_states(rnn, wd, torch.tensor([5, 4, 3, 2, 1, 0]))
    # the reference raises the same two for the same inputs
    for bad in ([3, 5, 2, 2, 1, 1], [5, 4, 3, 2, 1, 0]):
        with pytest.raises(RuntimeError):
            torch.nn.utils.rnn.pack_padded_sequence(words, torch.tensor(bad), batch_first=True)


def test_unsupported_gru_stays_on_the_library():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    dev = torch.device('cuda:0')
    x = torch.randn(2, 3, 300, device=dev)
    assert not ops.gru_last_supported(torch.nn.GRU(300, 1024, bidirectional=True, batch_first=True).to(dev), x)    # width not built
    assert not ops.gru_last_supported(torch.nn.GRU(300, 130, bidirectional=True, batch_first=True).to(dev), x)     # not a multiple of 4
    assert ops.gru_last_supported(torch.nn.GRU(300, 256, bidirectional=True, batch_first=True).to(dev), x)         # streamed weights
    assert not ops.gru_last_supported(torch.nn.GRU(300, 128, bidirectional=False, batch_first=True).to(dev), x)
    assert not ops.gru_last_supported(torch.nn.GRU(300, 128, bidirectional=True, batch_first=True), x.cpu())
    assert ops.gru_last_supported(torch.nn.GRU(300, 128, bidirectional=True, batch_first=True).to(dev), x)


def test_text_client_encoder_trains_the_same_on_both_recurrences():
    """EncoderText (bi-GRU + PIE) of a text client: 20 SGD steps on the fused recurrence and on the library's packed GRU from one
    initial state end in the same weights (2e-4 of scale) -- the trained outcome of the switch, not only one gradient."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    from creamfl_amd.networks.language_model import EncoderText
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    tokens = torch.randint(1, 500, (32, 20), generator=g)
    lengths = torch.tensor(sorted(torch.randint(1, 21, (32,), generator=g).tolist(), reverse=True))
    target = torch.nn.functional.normalize(torch.randn(32, 256, generator=g), dim=1).to(dev)
    torch.manual_seed(0)
    base = EncoderText(embed_dim=256, vocab_size=500)
    ends = []
    for fused in (True, False):
        m = EncoderText(embed_dim=256, vocab_size=500)
        m.load_state_dict(base.state_dict())
        m = m.to(dev).train()
        m.is_train = False
        opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        old = ops.GRU_FUSED[0]
        ops.GRU_FUSED[0] = fused
        try:
            for _ in range(20):
                opt.zero_grad()
                loss = (m(tokens.to(dev), lengths) - target).square().sum()
                loss.backward()
                opt.step()
        finally:
            ops.GRU_FUSED[0] = old
        ends.append(({k: v.detach().cpu() for k, v in m.state_dict().items()}, float(loss.detach())))
    (a, la), (b, lb) = ends
    assert abs(la - lb) <= 1e-4 * max(abs(lb), 1.0)
    for k in a:
        scale = float(b[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 2e-4 * max(scale, 1e-3), k


@pytest.mark.parametrize('shape', [(16, 12), (128, 32)])                # 192 and 4096 indices: both sides of torch's 3072 switch
def test_embedding_lookup_matches_nn_embedding(shape):
    """ops.embedding_lookup (index_select forward, index_add_ backward: one launch, capturable) against nn.Embedding."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(shape[0])
    emb = torch.nn.Embedding(11755, 300).to(dev)
    tokens = torch.randint(0, 40, shape, generator=g).to(dev)          # few distinct tokens: many rows collide
    gy = torch.randn(*shape, 300, generator=g).to(dev)
    a = ops.embedding_lookup(emb, tokens)
    (a * gy).sum().backward()
    ga = emb.weight.grad.clone()
    emb.weight.grad = None
    b = emb(tokens)
    (b * gy).sum().backward()
    assert torch.equal(a, b)
    scale = float(emb.weight.grad.abs().max())
    assert float((ga - emb.weight.grad).abs().max()) <= 1e-5 * scale        # (atomic adds: the order of the colliding rows differs)
    assert not ops.embedding_lookup(torch.nn.Embedding(10, 4, padding_idx=0).to(dev), tokens[:1] % 10).grad_fn.__class__.__name__.startswith('_EmbeddingFn')
