"""Where does a process spend its first minute?  Wall time of: import torch, engine build, first / second / third server step,
10 steady steps -- under the MIOPEN_FIND_MODE of the environment (runtime.py's default: 2).  One JSON line.
    MIOPEN_FIND_MODE=3 python tools/startup_probe.py [--cnn resnet101] [--batch 256]"""
import argparse
import json
import os
import sys
import time

t00 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cnn', default='resnet101')
ap.add_argument('--bert', default=None)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--dim', type=int, default=512)
args = ap.parse_args()
t = {'import_s': time.perf_counter() - t00}
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine  # noqa: E402
from creamfl_amd.utils.config import default_config  # noqa: E402
from creamfl_amd.utils.synthetic import coco_batch  # noqa: E402

dev = torch.device('cuda', 0)
t0 = time.perf_counter()
torch.manual_seed(0)
cfg = default_config(embed_dim=args.dim, cnn_type=args.cnn, not_bert=False)
if args.bert:
    cfg.model.bert_name = args.bert
eng = TrainerEngine(device=dev)
eng.create(cfg, {'<pad>': 0}, None, False)
eng.model_to_device()
eng.to_half()
eng.model.train()
b = coco_batch(args.batch, dev, seed=1, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)
torch.cuda.synchronize()
t['build_s'] = time.perf_counter() - t0
for k in ('step1_s', 'step2_s', 'step3_s'):
    t0 = time.perf_counter()
    eng.train_step(images, b[1], b[2], b[3])
    torch.cuda.synchronize()
    t[k] = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(10):
    eng.train_step(images, b[1], b[2], b[3])
torch.cuda.synchronize()
t['steady_ms_per_step'] = (time.perf_counter() - t0) * 100
t['find_mode'] = os.environ.get('MIOPEN_FIND_MODE')
t['total_s'] = time.perf_counter() - t00
from creamfl_amd import ops as _ops
t['find_db'] = {'covered_problems': _ops._FDB['hits'], 'uncovered_problems': _ops._FDB['misses'], 'auto': _ops._FDB['on']}
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()}))
