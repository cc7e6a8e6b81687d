#!/usr/bin/env python3
"""Is the bench step host-bound?  Per-step HOST time of `TrainerEngine.train_step` (time until the call returns, nothing waits for
the GPU inside the loop) next to the synchronised wall time per step.  Host time well below the wall time = the CPU runs ahead and
the GPU queue never drains; host time = wall time = the step is waiting for the CPU (or throttled by the queue depth)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401
import torch  # noqa: E402


def main():
    from creamfl_amd import _lib
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    _lib.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(256, dev, seed=1234, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        eng.train_step(images, b[1], b[2], b[3])
    torch.cuda.synchronize()
    n = 40
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        eng.train_step(images, b[1], b[2], b[3])
        host.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    host.sort()
    print(json.dumps({'what': 'host time per train_step call vs synchronised wall per step (config[1])',
                      'host_ms_median': round(host[n // 2] * 1e3, 2), 'host_ms_min': round(host[0] * 1e3, 2),
                      'host_ms_max': round(host[-1] * 1e3, 2), 'enqueue_all_ms_per_step': round(t_enq / n * 1e3, 2),
                      'wall_ms_per_step': round(wall * 1e3, 2)}))


if __name__ == '__main__':
    main()
