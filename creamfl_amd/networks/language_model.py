"""Client text encoder (row A2c).  Mirrors src/networks/language_model.py:28-130 (EncoderText): embedding ->
bi-GRU -> PIENet (HIP head) -> x scale -> ReLU -> class heads (train) | l2-normalised embedding (eval).
The reference reads the 11 755-word COCO vocabulary from a pickle inside its tree; here only its size is needed."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .. import ops
from .models.caption_encoder import get_pad_mask
from .models.pie_model import PIENet

COCO_VOCAB_SIZE = 11755


class EncoderText(nn.Module):
    def __init__(self, wemb_type=None, word_dim=300, embed_dim=2048, num_class=4, scale=128, mlp_local=False,
                 vocab_size=COCO_VOCAB_SIZE):
        super().__init__()
        self.embed_dim = embed_dim
        self.embed = nn.Embedding(vocab_size, word_dim)
        self.rnn = nn.GRU(word_dim, embed_dim // 2, bidirectional=True, batch_first=True)
        self.pie_net = PIENet(1, word_dim, embed_dim, word_dim // 2)
        self.relu = nn.ReLU(inplace=False)
        self.class_fc = nn.Linear(embed_dim, num_class)
        self.class_fc_2 = nn.Linear(embed_dim, 80)
        if wemb_type is not None:
            raise NotImplementedError('GloVe/FastText vectors need torchtext + a download; use wemb_type=None')
        nn.init.xavier_uniform_(self.embed.weight)
        self.is_train = True
        self.phase = ''
        self.scale = scale
        self.mlp_local = mlp_local
        if self.mlp_local:
            self.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                           nn.Linear(512, 512))

    def forward(self, x, lengths):
        lengths = lengths.cpu()
        wemb_out = self.embed(x)
        packed = pack_padded_sequence(wemb_out, lengths, batch_first=True)
        rnn_out, _ = self.rnn(packed)
        padded = pad_packed_sequence(rnn_out, batch_first=True)
        I = lengths.expand(self.embed_dim, 1, -1).permute(2, 1, 0) - 1
        out = torch.gather(padded[0], 1, I.to(x.device)).squeeze(1)
        pad_mask = get_pad_mask(wemb_out.shape[1], lengths, True)
        out, attn, residual = self.pie_net(out, wemb_out, pad_mask.to(out.device))
        out = out * self.scale
        out = self.relu(out)
        if self.is_train:
            fc_weight_relu = self.relu(self.class_fc.weight)
            self.class_fc.weight.data = fc_weight_relu
            x = self.class_fc(out)
            fc_weight_relu2 = self.relu(self.class_fc_2.weight)
            self.class_fc_2.weight.data = fc_weight_relu2
            x2 = self.class_fc_2(out)
            return x, x2, fc_weight_relu, fc_weight_relu2
        if self.mlp_local:
            out = self.head_proj(out)
        return ops.l2_normalize(out)
