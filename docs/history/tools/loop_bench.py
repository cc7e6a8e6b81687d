"""The PRODUCT loop next to the bench step (missing #3): pairs/s of `TrainerEngine.train(loader)` -- the loop
`MMFL.train` runs for the global contrastive phase (src/algorithms/retrieval_trainer.py:185-214, MMFL.py:188) -- fed from
HOST-resident batches, against `bench.py`'s bare `train_step` on one device-resident batch, in ONE process (same engine, same
clocks, same library choices).

    python tools/loop_bench.py [--steps 50] [--warmup 10] [--batch 256] [--pinned 1]

The loader yields the reference's batch tuples (src/datasets/_dataloader.py:49-64) from `--distinct` pre-generated host batches
(pinned like a `DataLoader(pin_memory=True)`'s, or pageable with --pinned 0: then the prefetch thread pins them); the 154 MB of
fp32 images per batch cross PCIe on the copy stream one batch ahead (creamfl_amd/utils/prefetch.py).  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401  (library set-up before torch initialises HIP)
import torch  # noqa: E402


class HostBatchLoader:
    """`steps` batches cycling over `distinct` host-resident ones."""

    def __init__(self, batches, steps):
        self.batches, self.steps = batches, steps
        self.dataset = None

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dim', type=int, default=512)
    ap.add_argument('--cnn', default='resnet101')
    ap.add_argument('--distinct', type=int, default=4)
    ap.add_argument('--pinned', type=int, default=1)
    ap.add_argument('--rounds', type=int, default=2, help='alternating (resident, loop) measurement rounds')
    args = ap.parse_args()
    from creamfl_amd import _lib
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    _lib.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=args.dim, cnn_type=args.cnn, not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()

    host = []
    for k in range(args.distinct):
        b = coco_batch(args.batch, 'cpu', seed=1234 + k, bert=True)
        if args.pinned:
            b = tuple(t.pin_memory() if torch.is_tensor(t) else t for t in b)
        host.append(b)
    res = coco_batch(args.batch, dev, seed=1234, bert=True)
    images = res[0].contiguous(memory_format=torch.channels_last)

    def resident(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.train_step(images, res[1], res[2], res[3])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def loop(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train(HostBatchLoader(host, n))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    resident(args.warmup)
    loop(args.warmup)
    r_ms, l_ms = [], []
    for _ in range(args.rounds):
        r_ms.append(resident(args.steps))
        l_ms.append(loop(args.steps))
    r, l = min(r_ms), min(l_ms)
    print(json.dumps({
        'what': 'TrainerEngine.train(loader) from host-resident batches vs bench.py-style train_step on a device-resident batch, same process',
        'config': {'cnn': args.cnn, 'text': 'bert-base', 'dim': args.dim, 'batch': args.batch, 'steps': args.steps,
                   'host_batches': 'pinned' if args.pinned else 'pageable (pinned by the prefetch thread)', 'distinct': args.distinct,
                   'h2d_bytes_per_step': int(sum(t.numel() * t.element_size() for t in host[0] if torch.is_tensor(t)))},
        'resident_ms_per_step': [round(v, 3) for v in r_ms], 'loop_ms_per_step': [round(v, 3) for v in l_ms],
        'resident_pairs_per_s': round(args.batch / r * 1e3, 1), 'loop_pairs_per_s': round(args.batch / l * 1e3, 1),
        'loop_over_resident': round(r / l, 4)}))


if __name__ == '__main__':
    main()
