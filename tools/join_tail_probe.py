#!/usr/bin/env python3
"""How long does the main stream wait at the end-of-backward join for the auxiliary streams (text tower, deferred weight
gradients)?  Events on both sides of streams.join_into_current, server step of bench.py (ResNet-101 + BERT-base, batch 256)."""
import os, sys, json
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401  (MIOpen user db, hardware queues)
import torch
from creamfl_amd import streams
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch

dev = torch.device('cuda', 0)
torch.backends.cudnn.benchmark = True
torch.manual_seed(1234)
cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
eng = TrainerEngine(device=dev); eng.create(cfg, {'<pad>': 0}, None, False); eng.model_to_device(); eng.to_half(); eng.model.train()
b = coco_batch(256, dev, seed=1234, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)
rec = []
orig = streams.join_into_current


def probe_join(device):
    cur = torch.cuda.current_stream(device)
    e_main = torch.cuda.Event(enable_timing=True); e_main.record(cur)
    ends = {}
    for (d, name), s in streams._STREAMS.items():
        if s != cur:
            e = torch.cuda.Event(enable_timing=True); e.record(s); ends[name] = e
    orig(device)
    e_after = torch.cuda.Event(enable_timing=True); e_after.record(cur)
    rec.append((e_main, ends, e_after))


streams.join_into_current = probe_join
for _ in range(5):
    eng.train_step(images, b[1], b[2], b[3])
torch.cuda.synchronize()
del rec[:]
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.train_step(images, b[1], b[2], b[3])
e1.record()
torch.cuda.synchronize()
out = {'ms_per_step': round(e0.elapsed_time(e1) / 10, 2), 'joins_per_step': len(rec) / 10, 'joins': []}
for (e_main, ends, e_after) in rec[-4:]:
    out['joins'].append({'main_waits_ms': round(e_main.elapsed_time(e_after), 3),
                         **{('%s_ends_after_main_ms' % k): round(e_main.elapsed_time(v), 3) for k, v in ends.items()}})
print(json.dumps(out, indent=1))
