"""Oracle for row A5: the con_w representation aggregation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/algorithms/MMFL.py:298-335 (closure `aggregation` inside
MMFL.distill; not importable in isolation, restated statement by statement).
"""
import torch


def conw_logprob(vec, global_other, literal=True, row_chunk=2048):
    """MMFL.py:304-307 for one client representation `vec` [M, D]:
        logits     = torch.matmul(vec, G_other.T)                     # [M, M]
        exp_logits = torch.exp(logits)                                 # no max-subtraction
        log_prob   = logits - torch.log(torch.sum(exp_logits, 1, keepdim=True))
        l          = torch.diagonal(log_prob)
    Row-chunked (rows are independent) so that M = 50 000 fits in memory; with
    literal=False uses logsumexp instead of log(sum(exp)).
    """
    m = vec.shape[0]
    out = torch.empty(m, dtype=vec.dtype)
    for r0 in range(0, m, row_chunk):
        r1 = min(m, r0 + row_chunk)
        logits = torch.matmul(vec[r0:r1], global_other.T)
        if literal:
            lse = torch.log(torch.sum(torch.exp(logits), dim=1))
        else:
            lse = torch.logsumexp(logits, dim=1)
        diag = logits[torch.arange(r1 - r0), torch.arange(r0, r1)]
        out[r0:r1] = diag - lse
    return out


def conw_weights(logprobs):
    """MMFL.py:311: contrastive_w = softmax over the client axis (dim 0) of
    the stacked [C, M] log-probs."""
    return torch.softmax(torch.stack(list(logprobs), 0), dim=0)


def conw_aggregate(vecs, global_other, literal=True):
    """MMFL.py:300-314 (image branch; the text branch :317-331 is the same with
    the roles of the two global banks swapped).  vecs: list of C tensors [M, D].
    Returns (agg [M, D], weights [C, M], logprobs [C, M]).
    """
    lps = [conw_logprob(v, global_other, literal) for v in vecs]
    w = conw_weights(lps)
    scaled = [(vecs[i] * w[i].reshape(-1, 1)).unsqueeze(0) for i in range(len(vecs))]
    agg = torch.sum(torch.cat(scaled, dim=0), dim=0)
    return agg, w, torch.stack(lps, 0)
