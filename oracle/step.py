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


# ---------------------------------------------------------------------------------------------------- client side
class ClientStepState:
    """An image client between two contrast steps, as plain tensors: the trainable state dict of its ResNet client net
    (resnet_client.py key names), the round-start copy the intra term compares against (ClientTrainer.py:195: deepcopy at the
    start of `run`), and SGD(lr, momentum 0.9, weight decay 5e-5) over the parameters (ClientTrainer.py:287)."""

    def __init__(self, state_dict, lr=1e-4, momentum=0.9, weight_decay=5e-5, layers=(2, 2, 2, 2)):
        self.sd = {k: v.detach().clone().float() if v.is_floating_point() else v.detach().clone() for k, v in state_dict.items()}
        self.old = {k: v.clone() for k, v in self.sd.items()}
        self.params = [k for k, v in self.sd.items() if v.is_floating_point() and 'running_' not in k]
        for k in self.params:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.SGD([self.sd[k] for k in self.params], lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.layers = layers


def client_contrast_step_cpu(state, images, global_img, global_txt, d_idx, interintra_weight=0.5, loss_scale=False):
    """One contrast step of an IMAGE client on the CPU (ClientTrainer.py:376-421): features of the model (train mode) and of
    the old model (eval mode, no grad) in the `extract_conv_feature` phase, inter term against the text bank + intra term
    against the image bank and the old features, backward, SGD step.  The `cpu_baseline` of bench.py --config 2."""
    from .bank_contrast import client_contrast_loss
    from .client_encoders import resnet_client_forward
    feat, new = resnet_client_forward(state.sd, images, 'extract_conv_feature', is_train=False, train_mode=True, layers=state.layers)
    with torch.no_grad():
        old, _ = resnet_client_forward(state.old, images, 'extract_conv_feature', is_train=False, train_mode=False, layers=state.layers)
    loss, _, _ = client_contrast_loss(feat, global_img, global_txt, d_idx, old, interintra_weight=interintra_weight,
                                      loss_scale=loss_scale)
    state.opt.zero_grad()
    loss.backward()
    state.opt.step()
    for k, v in new.items():
        if k in state.sd and not state.sd[k].requires_grad:
            state.sd[k] = v.detach()
    return loss.detach()
