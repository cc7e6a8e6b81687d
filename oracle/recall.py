"""Oracle for row A6: COCO retrieval recall (R@1/5/10, medr, meanr).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/algorithms/eval_coco.py
  recall_at_k              :22-29
  ParallelMatMulModule     :32-51   (mm, fold n_embeddings^2 copies, sort)
  extract_features         :118-195 (fp64 buffers [n, 7, D]; a [D] vector is
                                     broadcast into all 7 slots, :135,175,181)
  evaluate_recall          :273-334
"""
import numpy as np
import torch


def _replicate(features, n_embeddings):
    """extract_features :135-136,175,181: float64 buffer [n, n_embeddings, D]
    where every slot holds the same (fp32-valued) vector."""
    f = np.asarray(features, dtype=np.float64)
    return torch.from_numpy(np.repeat(f[:, None, :], n_embeddings, axis=1).copy())


def recall_ranks_literal(q_features, g_features, q_labels, g_labels, n_embeddings=7,
                         batch_size=1024):
    """evaluate_recall :273-317 + ParallelMatMulModule.forward :37-51, literally:
    fp64 mm of the 7x replicated vectors, 7x7 fold by summation, sort of -sims,
    then for each query the minimum sorted position over its positives.
    q_features [Nq, D], g_features [Ng, D] (any float dtype; values are widened
    to fp64 exactly as the reference's numpy buffers do).  Returns float64 ranks.
    """
    q = _replicate(q_features, n_embeddings)
    g = _replicate(g_features, n_embeddings)
    q_labels = np.asarray(q_labels)
    g_labels = np.asarray(g_labels)
    n_q, n_g = len(q_labels), len(g_labels)
    g_mat = g.view(n_g * n_embeddings, -1).t()
    best = np.zeros(n_q)
    for s in range(0, n_q, batch_size):
        q_idx = np.arange(s, min(n_q, s + batch_size))
        _q = q[q_idx, :].view(len(q_idx) * n_embeddings, -1)
        sims = _q.mm(g_mat)
        if n_embeddings > 1:
            sims = sims.view(len(q_idx), n_embeddings, n_g, n_embeddings)
            sims = sims.permute(0, 1, 3, 2)
            sims = torch.sum(torch.sum(sims, axis=1), axis=1)
        _, pred_ranks = (-sims).sort()
        for i, qi in enumerate(q_idx):
            pos = np.where(g_labels == q_labels[qi])[0]
            best[qi] = min(torch.where(pred_ranks[i] == p)[0][0].item() for p in pos)
    return best


def recall_ranks_count(q_features, g_features, q_labels, g_labels, block=2048):
    """Equivalent closed form (no ties): rank_q = #{g : sim(q,g) > max_{pos} sim(q,pos)},
    sims in fp64 = q . g (the 49x replication factor is a positive constant and
    cannot change an ordering)."""
    q = torch.from_numpy(np.asarray(q_features, dtype=np.float64))
    g = torch.from_numpy(np.asarray(g_features, dtype=np.float64))
    ql = torch.from_numpy(np.asarray(q_labels).astype(np.int64))
    gl = torch.from_numpy(np.asarray(g_labels).astype(np.int64))
    out = np.zeros(len(ql))
    for s in range(0, len(ql), block):
        sims = q[s:s + block] @ g.T
        posmask = ql[s:s + block, None] == gl[None, :]
        best = torch.where(posmask, sims, torch.full_like(sims, -np.inf)).max(1).values
        out[s:s + block] = (sims > best[:, None]).sum(1).numpy()
    return out


def recall_at_k(ranks, k):
    """eval_coco.py:22-29."""
    return 100.0 * len(np.where(ranks < k)[0]) / len(ranks)


def recall_scores(best_pred_ranks):
    """evaluate_recall :319-332."""
    r1 = recall_at_k(best_pred_ranks, 1)
    r5 = recall_at_k(best_pred_ranks, 5)
    r10 = recall_at_k(best_pred_ranks, 10)
    return {
        'recall_1': r1, 'recall_5': r5, 'recall_10': r10,
        'rsum': r1 + r5 + r10,
        'medr': np.floor(np.median(best_pred_ranks)) + 1,
        'meanr': np.mean(best_pred_ranks) + 1,
    }
