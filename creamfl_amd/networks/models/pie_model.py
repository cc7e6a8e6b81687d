"""Polysemous Instance Embedding (PIE) head on the HIP path.

Mirrors src/networks/models/pie_model.py (MultiHeadSelfAttention :11-40, PIENet :43-67): same
constructor arguments, parameter names (state_dict compatible) and return values.  The two dense
projections (w_1, fc) are library GEMMs; tanh -> w_2 dot -> masked softmax -> pooling and
sigmoid -> residual add -> LayerNorm (-> l2norm) are the hand-written kernels of csrc/pie.hip.
CreamFL always builds PIENet with n_embeds = 1 (image_encoder.py:34, caption_encoder.py:49);
other values are rejected.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


class MultiHeadSelfAttention(nn.Module):
    """Self-attention module by Lin, Zhouhan, et al. ICLR 2017 (n_head = 1)."""

    def __init__(self, n_head, d_in, d_hidden):
        super().__init__()
        if n_head != 1:
            raise NotImplementedError('creamfl_amd PIE head supports n_head == 1 (what CreamFL uses)')
        self.n_head = n_head
        self.w_1 = nn.Linear(d_in, d_hidden, bias=False)
        self.w_2 = nn.Linear(d_hidden, n_head, bias=False)
        self.init_weights()

    def init_weights(self):
        nn.init.xavier_uniform_(self.w_1.weight)
        nn.init.xavier_uniform_(self.w_2.weight)

    def pool(self, x, mask=None, want_mean=False):
        """x [b, seqlen, d_feat] -> (pooled [b, d_feat], attn [b, seqlen], xmean | empty)."""
        h = F.linear(x, self.w_1.weight)
        return ops.pie_pool(x, h, self.w_2.weight, mask, want_mean=want_mean)

    def forward(self, x, mask=None):
        output, attn, _ = self.pool(x, mask)
        return output, attn.unsqueeze(-1)          # attn [b, seqlen, n_head] as in the reference


class PIENet(nn.Module):
    """Polysemous Instance Embedding (PIE) module"""

    def __init__(self, n_embeds, d_in, d_out, d_h, dropout=0.0):
        super().__init__()
        if n_embeds != 1:
            raise NotImplementedError('creamfl_amd PIENet supports n_embeds == 1 (what CreamFL uses)')
        self.num_embeds = n_embeds
        self.attention = MultiHeadSelfAttention(n_embeds, d_in, d_h)
        self.fc = nn.Linear(d_in, d_out)
        self.sigmoid = nn.Sigmoid()
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_out)
        self.init_weights()

    def init_weights(self):
        nn.init.xavier_uniform_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0.0)

    def forward_fused(self, out, x, pad_mask=None, l2norm=False, out_from_mean=None):
        """One pass over x: attention pooling (+ mean pooling when `out_from_mean` is a module that maps
        the mean-pooled features to `out`, i.e. EncoderImage.fc), then the fused epilogue.
        Returns (y, o, attn [b, P, 1], residual) with y = l2norm(o) if l2norm else o."""
        pooled, attn, xmean = self.attention.pool(x, pad_mask, want_mean=out_from_mean is not None)
        if out_from_mean is not None:
            out = out_from_mean(xmean)
        if self.dropout.p > 0 and self.training:
            # the reference applies dropout to the sigmoid output (p = 0 everywhere in CreamFL)
            residual = self.dropout(torch.sigmoid(self.fc(pooled)))
            o = self.layer_norm(out + residual)
            return (ops.l2_normalize(o) if l2norm else o), o, attn.unsqueeze(-1), residual
        y, o, residual = ops.pie_epilogue(out, self.fc(pooled), self.layer_norm.weight, self.layer_norm.bias,
                                          self.layer_norm.eps, l2norm=l2norm)
        return y, o, attn.unsqueeze(-1), residual

    def forward(self, out, x, pad_mask=None):
        _, o, attn, residual = self.forward_fused(out, x, pad_mask, l2norm=False)
        return o, attn, residual
