"""CPU restatement of the text towers' recurrence, last valid step only (row A2c).  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py): imported by tests/ and nothing else.

Reference call sites: src/networks/language_model.py:93-107 and src/networks/models/caption_encoder.py:87-101 --
    packed  = pack_padded_sequence(embed(x), lengths, batch_first=True)
    padded  = pad_packed_sequence(rnn(packed)[0], batch_first=True)
    out     = gather(padded, 1, lengths - 1)                                  # [B, 2H]
with rnn = nn.GRU(word_dim, embed_dim // 2, bidirectional=True, batch_first=True).  The recurrence itself lives in a
third-party dependency (torch.nn.GRU of the reference's torch, here torch 2.10); its published definition
(torch.nn.GRU documentation, gate order r | z | n) is restated below in fp64 numpy, and `reference_formulation`
runs the reference's own three lines through torch on the CPU.  tests/test_oracle_golden.py pins the one against the
other, including the property the HIP path is built on: at position lengths - 1 the backward direction has taken
exactly ONE step (from a zero state, on the last word)."""
import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One step of torch.nn.GRU: r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r (W_hn h + b_hn)),
    h' = (1 - z) n + z h."""
    H = h.shape[-1]
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = _sigmoid(gi[..., :H] + gh[..., :H])
    z = _sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = np.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def bigru_last_states(words, lengths, params):
    """words [B, T, E], lengths [B] (>= 1), params = dict of the eight nn.GRU tensors (numpy) -> [B, 2H] in fp64:
    forward direction after lengths[b] steps | backward direction's first step (a cell on word lengths[b] - 1, zero state)."""
    words = np.asarray(words, np.float64)
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    B = words.shape[0]
    H = p['weight_hh_l0'].shape[1]
    out = np.zeros((B, 2 * H))
    for b in range(B):
        h = np.zeros(H)
        for t in range(int(lengths[b])):
            h = gru_cell(words[b, t], h, p['weight_ih_l0'], p['weight_hh_l0'], p['bias_ih_l0'], p['bias_hh_l0'])
        out[b, :H] = h
        out[b, H:] = gru_cell(words[b, int(lengths[b]) - 1], np.zeros(H), p['weight_ih_l0_reverse'], p['weight_hh_l0_reverse'],
                              p['bias_ih_l0_reverse'], p['bias_hh_l0_reverse'])
    return out


def reference_formulation(rnn, words, lengths):
    """The reference's own lines (language_model.py:99-107) on CPU tensors: differentiable, so tests take its gradients too."""
    lengths = lengths.cpu()
    states, _ = pad_packed_sequence(rnn(pack_padded_sequence(words, lengths, batch_first=True))[0], batch_first=True)
    last = (lengths - 1).view(-1, 1, 1).expand(-1, 1, states.shape[2])
    return states.gather(1, last).squeeze(1)


def gru_params(rnn):
    return {k: v.detach().cpu().numpy() for k, v in rnn.named_parameters()}
