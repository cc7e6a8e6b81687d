"""Oracle for row A1: PCME soft-contrastive loss over all N^2 image/caption pairs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/criterions/probemb.py
  batchwise_cdist        :7-45
  soft_contrastive_nll   :48-86
  full_sampling          :171-183
  pairwise_sampling      :150-169
  _compute_loss          :185-208
  match_prob             :210-219
  forward                :221-256
"""
import math

import torch


def _cdist_pairs(anchors, candidates, eps=1e-6):
    """d[i, j] = sqrt(sum_k (anchors[i,k] - candidates[j,k])^2 + eps).

    probemb.py:42-45 evaluates exactly this on the N^2 gathered rows produced by
    full_sampling (:171-183, anchor index i outer, candidate index j inner); a
    broadcasted difference visits the same (i, j) pairs in the same row-major
    order without the python double loop.  Chunked over i to bound memory.
    """
    n = anchors.shape[0]
    out = torch.empty(n, candidates.shape[0], dtype=anchors.dtype)
    step = max(1, (1 << 24) // max(1, candidates.numel()))
    rows = []
    for i0 in range(0, n, step):
        diff = anchors[i0:i0 + step, None, :] - candidates[None, :, :]
        rows.append(torch.sqrt((diff ** 2).sum(-1) + eps))
    return torch.cat(rows, 0) if rows else out


def _soft_contrastive_nll(logit, matched):
    """probemb.py:82-86 with K = 1 (2-D inputs => one 'sample' per pair).

    -( (logit*m - logsumexp([logit, -logit])) .logsumexp(dim=1) ) + log(1)
    """
    logit = logit[:, None]
    matched = matched[:, None]
    inner = logit * matched - torch.stack((logit, -logit), dim=2).logsumexp(dim=2)
    return -(inner.logsumexp(dim=1)) + math.log(logit.size(1))


def _compute_loss(x, y, negative_scale, shift, eps):
    """probemb.py:185-208: positives (i == j) and negatives summed separately."""
    n = x.shape[0]
    d = _cdist_pairs(x, y, eps).reshape(-1)
    matched = -torch.ones(n, n, dtype=x.dtype)
    matched.fill_diagonal_(1.0)
    matched = matched.reshape(-1)
    logits = -negative_scale * d + shift
    idx = matched == 1
    loss_pos = _soft_contrastive_nll(logits[idx], matched[idx]).sum()
    idx = matched != 1
    loss_neg = _soft_contrastive_nll(logits[idx], matched[idx]).sum()
    return {'loss': loss_pos + loss_neg, 'pos_loss': loss_pos, 'neg_loss': loss_neg}


def pair_loss_literal(image_features, caption_features, negative_scale, shift, eps=1e-6):
    """Statement-by-statement restatement of MCSoftContrastiveLoss.forward
    (probemb.py:221-256) for the configuration CreamFL runs (uniform_lambda = 0,
    vib_beta = 0, 2-D features).  Differentiable (torch autograd) w.r.t. every
    tensor argument.  Returns (loss, loss_dict) with the reference's 11 keys.
    """
    i2t = _compute_loss(image_features, caption_features, negative_scale, shift, eps)
    t2i = _compute_loss(caption_features, image_features, negative_scale, shift, eps)
    loss = i2t['loss'] + t2i['loss']
    loss_dict = {
        'i2t_loss': i2t['loss'].item(), 't2i_loss': t2i['loss'].item(),
        'i2t_pos_loss': i2t['pos_loss'].item(), 'i2t_neg_loss': i2t['neg_loss'].item(),
        't2i_pos_loss': t2i['pos_loss'].item(), 't2i_neg_loss': t2i['neg_loss'].item(),
        'uniform_loss': 0, 'vib_loss': 0,
        'shift': float(shift.detach()) if torch.is_tensor(shift) else float(shift),
        'negative_scale': float(negative_scale.detach()) if torch.is_tensor(negative_scale) else float(negative_scale),
        'loss': loss.item(),
    }
    return loss, loss_dict


def pair_loss_closed_form(image_features, caption_features, negative_scale, shift,
                          eps=1e-6, dtype=torch.float64):
    """Closed form of the same loss, evaluated in `dtype` (fp64 by default).

    NLL_ij = logsumexp(s, -s) - m*s = softplus(-2*m*s) with s = -a*d_ij + b,
    m = +1 on the diagonal, -1 elsewhere (probemb.py:82-86 with K = 1).  The two
    directions visit the transposed pair set, so loss = 2 * sum_ij NLL_ij.
    Returns dict(loss, pos, neg) where pos/neg are the one-direction partial sums.
    """
    I = image_features.to(dtype)
    T = caption_features.to(dtype)
    a = float(negative_scale)
    b = float(shift)
    d = _cdist_pairs(I, T, eps)
    s = -a * d + b
    n = I.shape[0]
    m = -torch.ones(n, n, dtype=dtype)
    m.fill_diagonal_(1.0)
    nll = torch.nn.functional.softplus(-2.0 * m * s)
    pos = torch.diagonal(nll).sum()
    neg = nll.sum() - pos
    return {'loss': 2.0 * (pos + neg), 'pos': pos, 'neg': neg}


def pair_loss_grads_closed_form(image_features, caption_features, negative_scale, shift,
                                eps=1e-6, dtype=torch.float64):
    """Analytic gradients of `loss` (both directions) in `dtype`.

    dL/dd_ij = 4*a*m_ij*sigmoid(-2*m_ij*s_ij) =: w_ij ;  c_ij = w_ij / d_ij
    dL/dI_i  = I_i * sum_j c_ij - sum_j c_ij T_j
    dL/dT_j  = T_j * sum_i c_ij - sum_i c_ij I_i
    dL/da    = sum_ij  4*m*sigmoid(-2ms)*d ;  dL/db = -sum_ij 4*m*sigmoid(-2ms)
    """
    I = image_features.to(dtype)
    T = caption_features.to(dtype)
    a = float(negative_scale)
    b = float(shift)
    d = _cdist_pairs(I, T, eps)
    s = -a * d + b
    n = I.shape[0]
    m = -torch.ones(n, n, dtype=dtype)
    m.fill_diagonal_(1.0)
    g = 4.0 * m * torch.sigmoid(-2.0 * m * s)
    c = a * g / d
    dI = I * c.sum(1, keepdim=True) - c @ T
    dT = T * c.sum(0)[:, None] - c.t() @ I
    return {'dI': dI, 'dT': dT, 'da': (g * d).sum(), 'db': -g.sum()}


def match_prob(image_features, caption_features, negative_scale, shift, eps=1e-6):
    """probemb.py:210-219 for 2-D inputs of equal length N (row-wise pairs):
    p_n = exp(s)/(exp(s)+exp(-s)), s = -a*||x_n - y_n|| + b."""
    diff = image_features - caption_features
    d = torch.sqrt((diff ** 2).sum(-1) + eps)
    logits = -negative_scale * d + shift
    return torch.exp(logits) / (torch.exp(logits) + torch.exp(-logits))
