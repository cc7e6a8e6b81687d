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
    lens_dev = lens.to(device)
    lens_dev._cfl_host_lens = tuple(lens.tolist())        # the lengths as the loader knows them, on the host (PCME._pack_plan)
    return (images.to(device), captions.to(device), None, lens_dev, ann_ids, image_ids, index)


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


def coco_batch_on_device(batch, device, seed=1234, vocab=11755, min_len=8, max_len=24, img=224, index0=0, captions_per_image=1):
    """The batch of `coco_batch`, GENERATED ON THE GPU (its own generator stream: other values than the host generator's, same
    distribution and the same batch-tuple contract).  A public set of 50 000 pairs is 30 GB of fp32 images: producing 77 MB per
    batch with the host's `randn` (~0.1 s) would bound every loop of a round; on the device it is one small kernel."""
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.randn(batch, 3, img, img, generator=g, device=device)
    # the lengths are drawn on the HOST (a loader knows them there: src/datasets/_dataloader.py:49-64 builds cap_lengths from python
    # lists) and travel with their device copy -- no device read for the frame width, and the packed text tower can plan on them
    host_lens = torch.randint(min_len, max_len + 1, (batch,), generator=torch.Generator().manual_seed(seed)).sort(descending=True).values
    L = int(host_lens.max())
    lens = host_lens.to(device)
    lens._cfl_host_lens = tuple(host_lens.tolist())
    captions = torch.randint(4, vocab, (batch, L), generator=g, device=device)
    pos = torch.arange(L, device=device)[None]
    captions[:, 0] = 1                                                     # <start>
    captions = torch.where(pos == lens[:, None] - 1, torch.full_like(captions, 2), captions)      # <end>
    captions = torch.where(pos >= lens[:, None], torch.zeros_like(captions), captions)            # <pad>
    index = list(range(index0, index0 + batch))
    return (images, captions, None, lens, list(index), [i // captions_per_image for i in index], index)


class DeviceCocoLoader(SyntheticCocoLoader):
    """SyntheticCocoLoader whose batches are born in HBM (`coco_batch_on_device`); same `.dataset` attributes."""

    def __iter__(self):
        for b, i0 in enumerate(range(0, self.n, self.bs)):
            yield coco_batch_on_device(min(self.bs, self.n - i0), self.device, self.seed * 100003 + b, self.vocab, img=self.img,
                                       index0=i0, captions_per_image=self.cpi)


class DeviceClientLoader:
    """A client's private training set (ClientTrainer's batch contracts), born in HBM:
         kind 'img': (inputs [B, 3, H, W] f32, labels [B] i64)                      -- CIFAR-100-shaped (resized to `img`)
         kind 'txt': (token ids [B, L] i64, labels [B] i64, lengths [B] i64 sorted descending)   -- AG_NEWS-shaped"""

    def __init__(self, kind, n, batch_size, classes, device, seed=0, img=224, vocab=11755, min_len=8, max_len=24):
        self.kind, self.n, self.bs, self.classes, self.device = kind, n, batch_size, classes, torch.device(device)
        self.seed, self.img, self.vocab, self.min_len, self.max_len = seed, img, vocab, min_len, max_len

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        for b, i0 in enumerate(range(0, self.n, self.bs)):
            m = min(self.bs, self.n - i0)
            g = torch.Generator(device=self.device).manual_seed(self.seed * 100003 + b)
            labels = torch.randint(0, self.classes, (m,), generator=g, device=self.device)
            if self.kind == 'img':
                yield torch.randn(m, 3, self.img, self.img, generator=g, device=self.device), labels
            else:
                lens = torch.randint(self.min_len, self.max_len + 1, (m,), generator=g, device=self.device).sort(descending=True).values
                L = int(lens.max())
                tok = torch.randint(4, self.vocab, (m, L), generator=g, device=self.device)
                tok = torch.where(torch.arange(L, device=self.device)[None] >= lens[:, None], torch.zeros_like(tok), tok)
                yield tok, labels, lens
