"""Client text encoder (row A2c): word embedding -> bidirectional GRU -> PIE attention head (HIP) -> scale -> ReLU, then
either the two classifier heads of the supervised phase or the l2-normalised embedding the contrast / representation phases
read.  Behavioural contract = src/networks/language_model.py:28-130 (EncoderText); parameter names are the reference's so
that its checkpoints load with strict=True (tests/golden/a2c_txt_*.npz are produced by the reference's own forward).

Differences by design: the reference unpickles its 11 755-word COCO vocabulary from inside its tree only to size the
embedding table -- here the size is an argument; pretrained word vectors (torchtext + a download) are refused."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .. import ops
from .models.caption_encoder import get_pad_mask
from .models.pie_model import PIENet

COCO_VOCAB_SIZE = 11755


def local_projection_head(width=512):
    """--mlp_local head (hard-wired to 512 in the reference: language_model.py:60-66, resnet_client.py:131-137)."""
    return nn.Sequential(nn.Linear(width, width), nn.BatchNorm1d(width), nn.ReLU(inplace=True), nn.Linear(width, width))


def clamped_head(linear, x):
    """Classifier head whose weight is clamped at zero IN PLACE before use (language_model.py:115-124,
    resnet_client.py:192-200: `weight.data = relu(weight)` -- the clamp persists in the parameter, and the clamped tensor is
    also handed to the centre loss).  Returns (logits, clamped weight)."""
    w = torch.relu(linear.weight)
    linear.weight.data = w
    return linear(x), w


class EncoderText(nn.Module):
    def __init__(self, wemb_type=None, word_dim=300, embed_dim=2048, num_class=4, scale=128, mlp_local=False,
                 vocab_size=COCO_VOCAB_SIZE):
        super().__init__()
        if wemb_type is not None:
            raise NotImplementedError('GloVe/FastText vectors need torchtext + a download; use wemb_type=None')
        self.embed_dim, self.scale, self.mlp_local = embed_dim, scale, mlp_local
        self.is_train, self.phase = True, ''            # switched from outside (ClientTrainer.py:372-375)
        self.embed = nn.Embedding(vocab_size, word_dim)
        nn.init.xavier_uniform_(self.embed.weight)
        self.rnn = nn.GRU(word_dim, embed_dim // 2, bidirectional=True, batch_first=True)
        self.pie_net = PIENet(1, word_dim, embed_dim, word_dim // 2)
        self.relu = nn.ReLU(inplace=False)
        self.class_fc = nn.Linear(embed_dim, num_class)
        self.class_fc_2 = nn.Linear(embed_dim, 80)
        if mlp_local:
            self.head_proj = local_projection_head()

    def sentence_states(self, tokens, lengths):
        """(GRU output at each sentence's last valid step [B, embed_dim], word embeddings [B, L, word_dim])."""
        words = ops.embedding_lookup(self.embed, tokens)
        if ops.gru_last_supported(self.rnn, words):
            # gru.hip: the forward direction's final state and the backward direction's first step are all the reference keeps
            # of the packed bi-GRU's output; lengths may stay on the device (no host round trip, capturable)
            return ops.bigru_last_states(self.rnn, words, lengths), words
        lengths = lengths.cpu()
        states, _ = pad_packed_sequence(self.rnn(pack_padded_sequence(words, lengths, batch_first=True))[0], batch_first=True)
        last = (lengths.to(tokens.device) - 1).view(-1, 1, 1).expand(-1, 1, self.embed_dim)
        return states.gather(1, last).squeeze(1), words

    def forward(self, x, lengths):
        final, words = self.sentence_states(x, lengths)
        pooled, _, _ = self.pie_net(final, words, get_pad_mask(words.shape[1], lengths.to(final.device), True))
        feat = self.relu(pooled * self.scale)
        if self.is_train:
            logits, w = clamped_head(self.class_fc, feat)
            logits2, w2 = clamped_head(self.class_fc_2, feat)
            return logits, logits2, w, w2
        return ops.l2_normalize(self.head_proj(feat) if self.mlp_local else feat)
