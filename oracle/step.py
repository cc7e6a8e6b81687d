"""CPU port of the server contrastive step (row S1) for the `cpu_baseline` leg of bench.py and for
end-to-end parity tests.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/algorithms/retrieval_trainer.py:192-214 (TrainerEngine.train body),
           src/networks/models/pcme.py:35-57, image_encoder.py:54-71, probemb.py:221-256.
The encoder trunks are the same plain torch.nn modules the product uses (they are library code below the
hot path); the PIE head, l2-normalise and the loss are evaluated with the oracle's restatements, never
with creamfl_amd ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pair_loss import pair_loss_literal
from .pie import pie_head, l2_normalize


def pcme_forward_cpu(model, images, sentences, lengths):
    """PCME.forward on CPU tensors with oracle head math.  `model` is a creamfl_amd PCME instance living on
    the CPU (only its parameters and torch.nn trunks are used)."""
    enc = model.img_enc
    fmap = enc.cnn.features(images)                                         # [N, Cd, 7, 7]
    n, cd = fmap.shape[0], fmap.shape[1]
    pooled = fmap.mean(dim=(2, 3)).view(-1, cd)                             # image_encoder.py:55
    out = enc.fc(pooled)
    x = fmap.view(-1, cd, fmap.shape[2] * fmap.shape[3]).transpose(1, 2)    # :62-64
    pn = enc.pie_net
    o, _, _ = pie_head(out, x, pn.attention.w_1.weight, pn.attention.w_2.weight, pn.fc.weight, pn.fc.bias,
                       pn.layer_norm.weight, pn.layer_norm.bias, ln_eps=pn.layer_norm.eps)
    img = l2_normalize(o)
    if model.config.not_bert:
        raise NotImplementedError('cpu port covers the BERT text tower (config 2)')
    hidden = model.txt_enc(**model._bert_inputs(sentences, None, lengths))['last_hidden_state']
    txt = l2_normalize(model.linear(hidden[:, 0, :]))
    return img, txt


def contrastive_step_cpu(model, criterion, optimizer, batch, grad_clip=2.0):
    """fwd -> MCSoftContrastiveLoss -> zero_grad -> backward -> clip_grad_norm_ -> optimizer.step."""
    images, captions, _, lens = batch[0], batch[1], batch[2], batch[3]
    img, txt = pcme_forward_cpu(model, images, captions, lens)
    loss, loss_dict = pair_loss_literal(img, txt, criterion.negative_scale, criterion.shift)
    optimizer.zero_grad()
    loss.backward()
    if grad_clip > 0:
        nn.utils.clip_grad.clip_grad_norm_(model.parameters(), grad_clip)
    optimizer.step()
    return loss.detach(), loss_dict
