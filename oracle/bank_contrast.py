"""Oracle for rows A3 (inter-modal contrast against the frozen global bank) and
A4 (intra-modal MOON-style contrast).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference (inline loop bodies, not importable in isolation):
  src/algorithms/ClientTrainer.py:369-429   both terms (uni-modal client)
  src/algorithms/ClientTrainer.py:431-480   intra only
  src/algorithms/ClientTrainer.py:482-507   inter only
  src/algorithms/MMClientTrainer.py:150-224 both terms (multi-modal client), :225-293 intra only, :294-324 inter only
The criterion is nn.CrossEntropyLoss (src/losses/__init__.py:19, mean reduction).
"""
import torch
import torch.nn.functional as F


def inter_contrast(feature, global_other, d_idx, temperature=0.5):
    """ClientTrainer.py:388 + :400-401 (MMClientTrainer.py:194-201):
        logits_inter = torch.div(torch.matmul(f, G_other.T), 0.5)
        loss_inter   = CrossEntropyLoss()(logits_inter, tensor(d_idx))
    """
    logits = torch.div(torch.matmul(feature, global_other.T), temperature)
    labels = torch.as_tensor(d_idx, dtype=torch.long)
    return F.cross_entropy(logits, labels)


def intra_contrast(feature, global_same, d_idx, old_feature, temperature=0.5):
    """ClientTrainer.py:386,404-414 (MMClientTrainer.py:170-191 stacks two
    modalities along dim 0 before the CE; pass concatenated tensors for that):
        target = G_same[d_idx]; pos = sum(f*target,-1); neg = sum(f*f_old,-1)
        logits = cat((pos, neg), 1) / 0.5 ; labels = 0
    """
    idx = torch.as_tensor(d_idx, dtype=torch.long)
    target = global_same[idx, :].type_as(feature)
    pos = torch.sum(feature * target, dim=-1).reshape(-1, 1)
    neg = torch.sum(feature * old_feature, dim=-1)
    logits = torch.cat((pos, neg.reshape(-1, 1)), dim=1)
    logits = logits / temperature
    labels = torch.zeros(feature.size(0), dtype=torch.long)
    return F.cross_entropy(logits, labels)


def client_contrast_loss(feature, global_same, global_other, d_idx, old_feature,
                         interintra_weight=0.5, loss_scale=False, use_inter=True,
                         use_intra=True, temperature=0.5):
    """The three flag combinations of ClientTrainer.tra (:369 / :431 / :482).

    both : (loss_moon + loss_inter) * w                         (:417)
           (loss_moon + loss_inter/(loss_inter/loss_moon).detach()) * w   (:419, --loss_scale)
    intra only: loss_moon (no weight, :470)     inter only: loss_inter (no weight, :502)
    Returns (loss, loss_inter|None, loss_moon|None).
    """
    loss_inter = inter_contrast(feature, global_other, d_idx, temperature) if use_inter else None
    loss_moon = intra_contrast(feature, global_same, d_idx, old_feature, temperature) if use_intra else None
    if use_inter and use_intra:
        if not loss_scale:
            loss = (loss_moon + loss_inter) * interintra_weight
        else:
            loss = (loss_moon + loss_inter / (loss_inter / loss_moon).detach()) * interintra_weight
    elif use_intra:
        loss = loss_moon
    elif use_inter:
        loss = loss_inter
    else:
        raise ValueError('no contrast term selected')
    return loss, loss_inter, loss_moon


def mm_client_contrast_loss(out_img, out_txt, global_img, global_txt, d_idx,
                            old_img=None, old_txt=None, interintra_weight=0.5, loss_scale=False,
                            use_inter=True, use_intra=True, temperature=0.5):
    """The three flag branches of MMClientTrainer.train_epoch:
      both  (MMClientTrainer.py:164-206): intra = CE over the stacked [2B, 2] logits (mean over 2B rows),
            inter = CE(img vs G_txt) + CE(txt vs G_img); (intra + inter) * w, or the --loss_scale form (:203-206)
      intra only (:246-264): the stacked CE alone, unweighted
      inter only (:301-308): loss_1 + loss_2, unweighted
    Returns (loss, loss_inter | None, loss_intra | None)."""
    loss_inter = loss_intra = None
    if use_intra:
        idx = torch.as_tensor(d_idx, dtype=torch.long)
        pos_i = torch.sum(out_img * global_img[idx].type_as(out_img), dim=-1).reshape(-1, 1)
        pos_t = torch.sum(out_txt * global_txt[idx].type_as(out_txt), dim=-1).reshape(-1, 1)
        neg_i = torch.sum(out_img * old_img, dim=-1)
        neg_t = torch.sum(out_txt * old_txt, dim=-1)
        logits_1 = torch.cat((pos_i, neg_i.reshape(-1, 1)), dim=1)
        logits_2 = torch.cat((pos_t, neg_t.reshape(-1, 1)), dim=1)
        logits = torch.cat((logits_1, logits_2), dim=0) / temperature
        labels = torch.zeros(out_img.size(0) * 2, dtype=torch.long)
        loss_intra = F.cross_entropy(logits, labels)
    if use_inter:
        loss_inter = (inter_contrast(out_img, global_txt, d_idx, temperature)
                      + inter_contrast(out_txt, global_img, d_idx, temperature))
    if use_inter and use_intra:
        if not loss_scale:
            loss = (loss_intra + loss_inter) * interintra_weight
        else:
            loss = (loss_intra + loss_inter / (loss_inter / loss_intra).detach()) * interintra_weight
    elif use_intra:
        loss = loss_intra
    elif use_inter:
        loss = loss_inter
    else:
        raise ValueError('no contrast term selected')
    return loss, loss_inter, loss_intra


def client_contrast_grads_closed_form(feature, global_same, global_other, d_idx, old_feature,
                                      temperature=0.5, dtype=torch.float64):
    """Closed forms in `dtype`:
        loss_inter = mean_b [ LSE_m(f_b.G_m / tau) - f_b.G_idx[b] / tau ]
        d/df_b     = (softmax_b - onehot_b) @ G / (tau * B)
        loss_moon  = mean_b softplus((neg_b - pos_b)/tau)
        d/df_b     = sigmoid((neg-pos)/tau) * (f_old_b - G_same[idx_b]) / (tau * B)
    Returns dict(loss_inter, loss_moon, lse, pos_inter, d_inter, d_moon).
    """
    f = feature.to(dtype)
    Go = global_other.to(dtype)
    Gs = global_same.to(dtype)
    fo = old_feature.to(dtype)
    idx = torch.as_tensor(d_idx, dtype=torch.long)
    B = f.shape[0]
    logits = (f @ Go.T) / temperature
    lse = torch.logsumexp(logits, dim=1)
    pos_inter = logits[torch.arange(B), idx]
    loss_inter = (lse - pos_inter).mean()
    p = torch.exp(logits - lse[:, None])
    p[torch.arange(B), idx] -= 1.0
    d_inter = (p @ Go) / (temperature * B)
    pos = (f * Gs[idx]).sum(-1)
    neg = (f * fo).sum(-1)
    z = (neg - pos) / temperature
    loss_moon = F.softplus(z).mean()
    d_moon = torch.sigmoid(z)[:, None] * (fo - Gs[idx]) / (temperature * B)
    return {'loss_inter': loss_inter, 'loss_moon': loss_moon, 'lse': lse,
            'pos_inter': pos_inter, 'd_inter': d_inter, 'd_moon': d_moon}
