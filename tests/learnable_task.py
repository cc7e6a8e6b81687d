"""A synthetic retrieval task a model can LEARN, for training-outcome checks (test infrastructure).

There is no dataset offline, and random images / random captions carry no signal: a model trained on them ends at chance whatever
its kernels do.  Here every identity i has
  * an image prototype: a seeded 3 x 8 x 8 pattern, nearest-upsampled to S x S; a sample = prototype + 0.5 * noise;
  * a caption signature: <start>, the three base-16 digits of i as three tokens of three disjoint ranges, then random filler
    words from a fourth range (count 2 .. 6, fresh per sample), <end>.
A training batch holds B DISTINCT identities (no false negatives inside the all-pairs loss, src/criterions/probemb.py:171-183);
the held-out evaluation set is n_eval identities x 5 captions with fresh noise and fresh fillers, in the batch-tuple contract of
src/datasets/_dataloader.py:49-64 (every caption row repeats its image), with the `.dataset` attributes COCOEvaluator reads.
Chance R@1 is 100 / n_eval.
`caption_swap` = p > 0 makes the task AMBIGUOUS instead of merely longer to learn: every caption (training and held-out) carries,
with probability p, the signature of a random other identity.  A converged model then ends near R@1 = 100 (1 - p) in both
directions whatever its precision -- a ceiling that is set by the data, reached on a plateau (stable across seeds), and well below
100 %, so that a path whose gradients are systematically off (it fits the clean pairs less sharply, or the false ones more) shows up
as points of R@1 instead of disappearing into 99 %."""
import torch


class LearnableTask:
    def __init__(self, n_id=1000, img=64, seed=0, noise=0.5, device='cpu', caption_swap=0.0):
        g = torch.Generator().manual_seed(seed)
        self.n_id, self.img, self.noise, self.device = n_id, img, noise, torch.device(device)
        self.caption_swap = float(caption_swap)
        proto = torch.randn(n_id, 3, 8, 8, generator=g)
        self.proto = torch.nn.functional.interpolate(proto, size=(img, img), mode='nearest').to(self.device)

    def _captions(self, ids, gen):
        B = len(ids)
        if self.caption_swap > 0:
            swap = torch.rand(B, generator=gen) < self.caption_swap
            ids = torch.where(swap, torch.randint(0, self.n_id, (B,), generator=gen), ids)
        nf = torch.randint(2, 7, (B,), generator=gen)
        lens = nf + 5                                              # <start> + 3 digits + fillers + <end>
        L = int(lens.max())
        cap = torch.zeros(B, L, dtype=torch.int64)
        cap[:, 0] = 1
        for k in range(3):
            cap[:, 1 + k] = 100 + 16 * k + ((ids >> (4 * k)) & 15)
        fill = torch.randint(1000, 3000, (B, L), generator=gen)
        pos = torch.arange(L)[None]
        body = (pos >= 4) & (pos < (lens - 1)[:, None])
        cap = torch.where(body, fill, cap)
        cap[torch.arange(B), lens - 1] = 2
        return cap, lens

    def batch(self, ids, seed):
        """(images, captions, None, lens) for the identities `ids` (int64 tensor), sorted by caption length descending."""
        gen = torch.Generator().manual_seed(seed)
        cap, lens = self._captions(ids, gen)
        order = torch.argsort(lens, descending=True, stable=True)
        ids, cap, lens = ids[order], cap[order], lens[order]
        gd = torch.Generator(device=self.device).manual_seed(seed) if self.device.type == 'cuda' else gen
        images = self.proto[ids.to(self.device)] + self.noise * torch.randn(len(ids), 3, self.img, self.img, generator=gd,
                                                                             device=self.device)
        return images, cap.to(self.device), None, lens.to(self.device), ids

    def train_batch(self, step, B):
        gen = torch.Generator().manual_seed(1000 + step)
        ids = torch.randperm(self.n_id, generator=gen)[:B]
        return self.batch(ids, 5000 + step)[:4]


class _EvalSet:
    iid_to_cls = {}

    def __init__(self, n_images, n_captions):
        self.n_images, self.n = n_images, n_captions

    def __len__(self):
        return self.n


class EvalLoader:
    """n_eval identities x 5 captions of fresh samples; every caption row carries its (noisy) image."""

    def __init__(self, task, n_eval=1000, per_image=5, images_per_batch=50, seed=777):
        self.task, self.n_eval, self.per, self.ipb, self.seed = task, n_eval, per_image, images_per_batch, seed
        self.dataset = _EvalSet(n_eval, n_eval * per_image)

    def __len__(self):
        return (self.n_eval + self.ipb - 1) // self.ipb

    def __iter__(self):
        for b, i0 in enumerate(range(0, self.n_eval, self.ipb)):
            ids_img = torch.arange(i0, min(self.n_eval, i0 + self.ipb))
            ids = ids_img.repeat_interleave(self.per)
            images, cap, _, lens, ids_sorted = self.task.batch(ids, self.seed + b)
            # rows were reordered by caption length: ids follow; annotation ids are unique per row
            ann = [int(i) * self.per + k for k, i in enumerate(ids_sorted.tolist())]
            # one image per identity in the evaluation: use the prototype sample of the FIRST row of each identity for all its rows
            first = {}
            for row, i in enumerate(ids_sorted.tolist()):
                first.setdefault(i, row)
            src = torch.as_tensor([first[i] for i in ids_sorted.tolist()], device=images.device)
            images = images[src]
            yield images, cap, None, lens, ann, ids_sorted.tolist(), ann
