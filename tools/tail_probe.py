#!/usr/bin/env python3
"""How long does the main stream WAIT for the side streams at the end of the backward pass of the bench step (configs[1])?
The weight gradients run on the 'wgrad' stream, flushed in groups when the main stream enters a residual BatchNorm backward and
once more at the end of the pass; then the main stream joins every side stream before the clip's gradient norm.  A kernel trace
cannot tell (the profiler serialises the queues: 56 ms per step instead of 42.7), HIP events can: per step, an event on the main
stream in front of the join, one at the tail of every side stream, one behind the join.
    python tools/tail_probe.py [--steps 20] [--flush N]      (GPU)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from creamfl_amd import runtime  # noqa: E402

runtime.configure_env()
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--flush', type=int, default=0, help='streams.FLUSH_POLICY')
    args = ap.parse_args()
    from creamfl_amd import streams
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(args.batch, dev, seed=1234, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)
    streams.FLUSH_POLICY[0] = args.flush
    rec = []
    orig_join, orig_flush = streams.join_into_current, streams.flush
    last_flush = [0]

    def flush(device, limit=None):
        last_flush[0] = len(streams._PENDING)
        return orig_flush(device, limit)

    def join(device):
        cur = torch.cuda.current_stream(device)
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        tails = {}
        for (d, name), s in streams._STREAMS.items():
            if d == torch.device(device) and s != cur:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(s)
                tails[name] = ev
        orig_join(device)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(cur)
        rec.append((e0, tails, e1, last_flush[0]))

    for _ in range(8):
        eng.train_step(images, b[1], b[2], b[3])
    torch.cuda.synchronize()
    streams.join_into_current, streams.flush = join, flush
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps):
        eng.train_step(images, b[1], b[2], b[3])
    t1.record()
    torch.cuda.synchronize()
    streams.join_into_current, streams.flush = orig_join, orig_flush
    wait = [e0.elapsed_time(e1) for e0, _, e1, _ in rec]
    out = {'steps': args.steps, 'flush_policy': args.flush, 'ms_per_step': round(t0.elapsed_time(t1) / args.steps, 3),
           'joins_per_step': len(rec) / args.steps,
           'main_waits_at_the_join_ms': round(sum(wait) / len(wait), 3), 'max_ms': round(max(wait), 3),
           'tasks_in_the_last_flush': round(sum(r[3] for r in rec) / len(rec), 1), 'side_stream_tail_after_main_ms': {}}
    for name in rec[0][1]:
        v = [r[0].elapsed_time(r[1][name]) for r in rec if name in r[1]]
        out['side_stream_tail_after_main_ms'][name] = round(sum(v) / len(v), 3)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
