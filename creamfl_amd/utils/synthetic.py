"""MSCOCO-shaped synthetic batches: there is no network for datasets, and the headline
metric is defined on synthetic inputs of the reference's batch-tuple contract
(src/datasets/_dataloader.py:49-64):
    (images [B,3,224,224] f32, captions [B,Lmax] i64 0-padded & sorted by length desc, captions_word,
     caption_lens [B] i64, ann_ids, image_ids, index)
"""
import torch


def coco_batch(batch, device='cpu', seed=1234, bert=True, vocab=11755, min_len=8, max_len=24, img=224,
               index0=0, captions_per_image=1):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, img, img, generator=g)
    lens = torch.randint(min_len, max_len + 1, (batch,), generator=g).sort(descending=True).values
    L = int(lens.max())
    # `captions` always holds COCO-vocabulary ids (<pad>=0 <start>=1 <end>=2 <unk>=3, 11 755 words), as the
    # reference's loaders do for both towers; PCME's BERT path maps them into BERT's id space when no
    # tokenizer is attached (`bert` is kept for call-site readability only).
    captions = torch.randint(4, vocab, (batch, L), generator=g)
    start, end = 1, 2
    captions[:, 0] = start
    for i, l in enumerate(lens.tolist()):
        captions[i, l - 1] = end
        captions[i, l:] = 0
    index = list(range(index0, index0 + batch))
    ann_ids = list(index)
    image_ids = [i // captions_per_image for i in index]
    return (images.to(device), captions.to(device), None, lens.to(device), ann_ids, image_ids, index)


class _Dataset:
    def __init__(self, n_captions, n_images):
        self.n = n_captions
        self.n_images = n_images
        self.iid_to_cls = {}

    def __len__(self):
        return self.n


class SyntheticCocoLoader:
    """Finite loader of synthetic batches with the `.dataset` attributes the evaluator and MMFL read
    (n_images, iid_to_cls, __len__)."""

    def __init__(self, n_pairs, batch_size, seed=0, bert=True, captions_per_image=1, device='cpu', img=224,
                 vocab=11755):
        self.n, self.bs, self.seed, self.bert, self.cpi = n_pairs, batch_size, seed, bert, captions_per_image
        self.device, self.img, self.vocab = device, img, vocab
        self.dataset = _Dataset(n_pairs, n_pairs // captions_per_image)

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        for b, i0 in enumerate(range(0, self.n, self.bs)):
            yield coco_batch(min(self.bs, self.n - i0), self.device, self.seed * 100003 + b, self.bert, self.vocab,
                             img=self.img, index0=i0, captions_per_image=self.cpi)
