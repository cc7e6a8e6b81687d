"""Oracle for row A2-head: the PIE attention-pooling head of PCME.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference:
  src/networks/models/pie_model.py:11-40   MultiHeadSelfAttention (n_head = 1)
  src/networks/models/pie_model.py:43-67   PIENet
  src/networks/models/image_encoder.py:54-71  EncoderImage.forward glue
  src/utils/tensor_utils.py:25-27          l2_normalize
Parameters are passed as plain tensors (names = the reference's state_dict keys).
"""
import torch
import torch.nn.functional as F


def l2_normalize(x, axis=-1):
    """tensor_utils.py:25-27: F.normalize(p=2) => x / max(||x||_2, 1e-12)."""
    return F.normalize(x, p=2, dim=axis)


def pie_attention_pool(x, w1, w2, pad_mask=None):
    """pie_model.py:28-40 with n_head = 1.
        attn = w_2(tanh(w_1(x)))                # [B, P, 1]
        attn.masked_fill_(mask, -inf)           # mask [B, P] True = padded
        attn = softmax(attn, dim=1)
        out  = bmm(attn^T, x).squeeze(1)        # [B, Cd]
    x [B, P, Cd]; w1 [dh, Cd]; w2 [1, dh].  Returns (pooled [B, Cd], attn [B, P, 1]).
    """
    attn = F.linear(torch.tanh(F.linear(x, w1)), w2)
    if pad_mask is not None:
        mask = pad_mask.repeat(1, 1, 1).permute(1, 2, 0)
        attn = attn.masked_fill(mask, float('-inf'))
    attn = torch.softmax(attn, dim=1)
    output = torch.bmm(attn.transpose(1, 2), x)
    if output.shape[1] == 1:
        output = output.squeeze(1)
    return output, attn


def pie_head(out, x, w1, w2, fc_w, fc_b, ln_w, ln_b, pad_mask=None, ln_eps=1e-5):
    """PIENet.forward, pie_model.py:61-67 (num_embeds = 1, dropout p = 0):
        residual, attn = attention(x, mask)
        residual = sigmoid(fc(residual))
        out = layer_norm(out + residual)
    Returns (out [B, D], attn [B, P, 1], residual [B, D]).
    """
    pooled, attn = pie_attention_pool(x, w1, w2, pad_mask)
    residual = torch.sigmoid(F.linear(pooled, fc_w, fc_b))
    out = F.layer_norm(out + residual, (out.shape[-1],), ln_w, ln_b, ln_eps)
    return out, attn, residual


def image_head_glue(out_7x7, fc_w, fc_b, pie_params):
    """image_encoder.py:54-71 after the CNN trunk (mlp_local = False):
        pooled = avgpool(out_7x7).view(-1, Cd); out = fc(pooled)
        out, attn, residual = pie_net(out, out_7x7.view(-1, Cd, 49).transpose(1, 2))
        out = l2_normalize(out)
    out_7x7 [B, Cd, 7, 7]; pie_params = dict(w1, w2, fc_w, fc_b, ln_w, ln_b).
    """
    b, cd = out_7x7.shape[0], out_7x7.shape[1]
    pooled = out_7x7.mean(dim=(2, 3)).view(-1, cd)
    out = F.linear(pooled, fc_w, fc_b)
    x = out_7x7.view(-1, cd, 49).transpose(1, 2)
    out, attn, residual = pie_head(out, x, **pie_params)
    return l2_normalize(out), attn, residual
