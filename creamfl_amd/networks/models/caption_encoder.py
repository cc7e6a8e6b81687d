"""GRU text tower.  Mirrors src/networks/models/caption_encoder.py:21-116 (get_pad_mask, EncoderText):
embedding -> bi-GRU (MIOpen) -> last valid step -> PIENet over the word embeddings with a pad mask ->
l2-normalise (BEFORE head_proj, the reference's own order, :109-112)."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from ... import ops
from .pie_model import PIENet


def get_pad_mask(max_length, lengths, set_pad_to_one=True):
    ind = torch.arange(0, max_length, device=lengths.device).unsqueeze(0)       # (born on the lengths' device: capturable)
    mask = (ind >= lengths.unsqueeze(1)) if set_pad_to_one else (ind < lengths.unsqueeze(1))
    return mask.to(lengths.device)


class EncoderText(nn.Module):
    def __init__(self, word2idx, opt, mlp_local):
        super().__init__()
        wemb_type, word_dim, embed_dim = opt.wemb_type, opt.word_dim, opt.embed_dim
        self.embed_dim = embed_dim
        self.embed = nn.Embedding(len(word2idx), word_dim)
        self.embed.weight.requires_grad = True
        self.rnn = nn.GRU(word_dim, embed_dim // 2, bidirectional=True, batch_first=True)
        self.pie_net = PIENet(1, word_dim, embed_dim, word_dim // 2)
        self.init_weights(wemb_type, word2idx, word_dim)
        self.n_samples_inference = opt.get('n_samples_inference', 0)
        self.mlp_local = mlp_local
        if self.mlp_local:
            self.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                           nn.Linear(512, 512))

    def init_weights(self, wemb_type, word2idx, word_dim, cache_dir=None):
        if wemb_type is None:
            nn.init.xavier_uniform_(self.embed.weight)
        else:
            raise NotImplementedError('pretrained GloVe/FastText vectors need torchtext + a download; '
                                      'construct with wemb_type=None and load embed.weight yourself')

    def forward(self, x, lengths):
        wemb_out = ops.embedding_lookup(self.embed, x)
        if ops.gru_last_supported(self.rnn, wemb_out):
            # gru.hip: [forward direction's final state | backward direction's first step] = the gather below, lengths on the device
            out = ops.bigru_last_states(self.rnn, wemb_out, lengths)
            pad_mask = get_pad_mask(wemb_out.shape[1], lengths.to(out.device), True)
        else:
            lengths = lengths.cpu()
            packed = pack_padded_sequence(wemb_out, lengths, batch_first=True)
            rnn_out, _ = self.rnn(packed)
            padded = pad_packed_sequence(rnn_out, batch_first=True)
            I = lengths.expand(self.embed_dim, 1, -1).permute(2, 1, 0) - 1
            out = torch.gather(padded[0], 1, I.to(x.device)).squeeze(1)
            pad_mask = get_pad_mask(wemb_out.shape[1], lengths, True).to(out.device)
        output = {}
        if not self.mlp_local:
            out, _, attn, residual = self.pie_net.forward_fused(out, wemb_out, pad_mask, l2norm=True)
        else:
            out, _, attn, residual = self.pie_net.forward_fused(out, wemb_out, pad_mask, l2norm=True)
            out = self.head_proj(out)
        output['embedding'] = out
        return output
