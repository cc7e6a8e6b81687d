#!/usr/bin/env python3
"""Where does the single-rank multi-GPU path lose time?  Variants of the server step on one GPU (10 steps each)."""
import os, sys, time, json
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
import torch.distributed as dist
from creamfl_amd import dist as cdist
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch

dev = torch.device('cuda', 0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.dup2(2, 1)
dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
torch.manual_seed(1234)
cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
eng = TrainerEngine(device=dev); eng.create(cfg, {'<pad>': 0}, None, False); eng.model_to_device(); eng.to_half(); eng.model.train()
b = coco_batch(256, dev, seed=1234, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)

def run(tag, n=10):
    for _ in range(3): eng.train_step(images, b[1], b[2], b[3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.train_step(images, b[1], b[2], b[3])
    torch.cuda.synchronize()
    sys.stderr.write(json.dumps({'variant': tag, 'ms_per_step': round((time.perf_counter() - t0) / n * 1e3, 2)}) + '\n')

if os.environ.get('DP_FIRST'):
    eng.enable_data_parallel()
    run('dp from the first step')
    run('dp from the first step (again)')
    dist.destroy_process_group()
    sys.exit(0)
run('plain')
eng.enable_data_parallel()
run('dp (GradBuckets)')
red = eng.dp.reducer
orig_reduce = red._reduce
red._reduce = lambda bi: None
run('dp, reduce disabled (hooks + notify only)')
red._reduce = orig_reduce
for h in red._hooks: h.remove()
run('dp, autograd hooks removed (only deferred notify)')
dist.destroy_process_group()
