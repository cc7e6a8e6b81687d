"""PCME two-tower wrapper.  Mirrors src/networks/models/pcme.py:15-63: same constructor, attributes
(img_enc, txt_enc, tokenizer, linear, n_embeddings, embed_dim) and the same 10-key output dict.

Text tower: `not_bert` -> GRU EncoderText; else BERT -> CLS -> Linear(768, D) -> l2-normalise (:40-44).
The reference tokenises `captions_word` with a downloaded BertTokenizer on every step; offline there is
no vocabulary, so when no tokenizer is attached the COCO-vocabulary `sentences` ids are mapped into BERT's id
space (`_bert_inputs`) and `lengths` gives the attention mask.  Attach a tokenizer with `model.tokenizer = ...` to get the
reference behaviour.
"""
import os

import torch
import torch.nn as nn

from ... import ops, streams
from ..backbones import BertModel
from .caption_encoder import EncoderText
from .image_encoder import EncoderImage


_NO_TWO_STREAM = bool(os.environ.get('CFL_NO_TWO_STREAM'))         # A/B switch for measurements
_NO_BERT_PACK = [bool(os.environ.get('CFL_NO_BERT_PACK'))]     # A/B: the BERT tower on the padded [B, L] frame (round 5's form)


class PCME(nn.Module):
    """Probabilistic CrossModal Embedding (PCME) module"""

    def __init__(self, word2idx, config, mlp_local):
        super().__init__()
        self.config = config
        self.embed_dim = config.embed_dim
        if config.get('n_samples_inference', 0):
            self.n_embeddings = config.n_samples_inference
        else:
            self.n_embeddings = 1
        self.img_enc = EncoderImage(config, mlp_local)
        if config.not_bert:
            self.txt_enc = EncoderText(word2idx, config, mlp_local)
        else:
            self.txt_enc = BertModel.from_pretrained(config.get('bert_name', 'bert-base-uncased'))
            self.tokenizer = None
            self.linear = nn.Linear(self.txt_enc.config.hidden_size, self.embed_dim)

    def _bert_inputs(self, sentences, captions_word, lengths):
        if getattr(self, 'tokenizer', None) is not None and captions_word is not None:
            inputs = self.tokenizer(captions_word, padding=True, return_tensors='pt')
            dev = self.linear.weight.device
            return {k: v.to(dev) for k, v in inputs.items()}
        # offline stand-in for the tokenizer: COCO-vocabulary ids -> BERT id space
        # (<pad> 0 -> [PAD] 0, <start> 1 -> [CLS] 101, <end> 2 -> [SEP] 102, word w -> 1000 + w)
        L = sentences.shape[1]
        mask = torch.arange(L, device=sentences.device)[None, :] < lengths.to(sentences.device)[:, None]
        ids = torch.where(sentences == 0, sentences,
                          torch.where(sentences == 1, torch.full_like(sentences, 101),
                                      torch.where(sentences == 2, torch.full_like(sentences, 102), sentences + 1000)))
        ids = ids.clamp_max(self.txt_enc.config.vocab_size - 1)
        out = {'input_ids': ids, 'attention_mask': mask}
        plan = self._pack_plan(lengths, L, sentences.device)
        if plan is not None:
            out['pack'] = plan
        return out

    def _pack_plan(self, lengths, L, device):
        """The packed-run plan of this batch (BertModel.pack_plan) when the lengths are known ON THE HOST -- a CPU tensor / list, or a
        device tensor that carries its host copy (`_cfl_host_lens`: utils/synthetic.coco_batch and utils/prefetch.DevicePrefetcher
        attach it); a bare device tensor gets the padded frame (reading it would stall the step's issue thread on the GPU).
        The plan of the last batch is kept: a resident batch (bench.py) builds it once.  CFL_NO_BERT_PACK=1 switches packing off."""
        if _NO_BERT_PACK[0] or not isinstance(self.txt_enc, BertModel) or torch.device(device).type != 'cuda':
            return None
        host = getattr(lengths, '_cfl_host_lens', None)
        if host is None:
            if torch.is_tensor(lengths):
                if lengths.is_cuda:
                    return None
                host = tuple(lengths.tolist())
            else:
                host = tuple(int(v) for v in lengths)
        c = getattr(self, '_pack_cache', None)
        if c is not None and (c[0] is host or c[0] == host) and c[1] == L and c[2] == device:
            return c[3]
        plan = BertModel.pack_plan(host, L, device)
        self._pack_cache = (host, L, device, plan)
        return plan

    def _text_tower(self, sentences, captions_word, lengths):
        if self.config.not_bert:
            return self.txt_enc(sentences, lengths)
        extra = {'cls_only': True} if isinstance(self.txt_enc, BertModel) else {}     # only [:, 0, :] is read
        hidden = self.txt_enc(**self._bert_inputs(sentences, captions_word, lengths), **extra)['last_hidden_state']
        return {'embedding': ops.l2_normalize(self.linear(hidden[:, 0, :]))}

    def forward(self, images, sentences, captions_word, lengths):
        # The two towers are independent until the loss.  On the GPU the text tower (many small, latency-bound kernels)
        # runs on a second HIP stream next to the image tower (large HBM-bound kernels); autograd replays each
        # backward op on the stream of its forward, so the two backward passes overlap as well.
        side = None
        if images.is_cuda and not _NO_TWO_STREAM and torch.is_tensor(sentences) and sentences.is_cuda:
            side = streams.get(images.device, 'text')
        if side is None:
            image_output = self.img_enc(images)
            caption_output = self._text_tower(sentences, captions_word, lengths)
        else:
            main = torch.cuda.current_stream(images.device)
            side.wait_stream(main)                          # the inputs (and the weights of the last step) are ready
            with torch.cuda.stream(side):
                caption_output = self._text_tower(sentences, captions_word, lengths)
            image_output = self.img_enc(images)
            main.wait_stream(side)
            for t in caption_output.values():               # allocated on the side stream, consumed on the main one
                if torch.is_tensor(t):
                    t.record_stream(main)
        return {
            'image_features': image_output['embedding'],
            'image_attentions': image_output.get('attention'),
            'image_residuals': image_output.get('residual'),
            'image_logsigma': image_output.get('logsigma'),
            'image_logsigma_att': image_output.get('uncertainty_attention'),
            'caption_features': caption_output['embedding'],
            'caption_attentions': caption_output.get('attention'),
            'caption_residuals': caption_output.get('residual'),
            'caption_logsigma': caption_output.get('logsigma'),
            'caption_logsigma_att': caption_output.get('uncertainty_attention'),
        }

    def image_forward(self, images):
        return self.img_enc(images)

    def text_forward(self, sentences, lengths):
        return self.txt_enc(sentences, lengths)
