"""Per-step wall time of the bench step (HIP events around every step, no host sync inside the loop): shows whether a slow
run is uniformly slow or a few long steps."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
from creamfl_amd import _lib
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch
_lib.load()
dev = torch.device('cuda', 0)
torch.backends.cudnn.benchmark = True
torch.manual_seed(1234)
cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
eng = TrainerEngine(device=dev)
eng.create(cfg, {'<pad>': 0}, None, False)
eng.model_to_device(); eng.to_half(); eng.model.train()
b = coco_batch(256, dev, seed=1234, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
if os.environ.get('JIT_NOGC'):
    import gc
    gc.collect(); gc.freeze(); gc.disable()
if os.environ.get('JIT_RESERVE_GB'):
    from creamfl_amd import streams
    gb = int(os.environ['JIT_RESERVE_GB'])
    keep = [torch.empty(gb << 30, dtype=torch.uint8, device=dev)]
    for nm in ('text', 'wgrad'):
        with torch.cuda.stream(streams.get(dev, nm)):
            keep.append(torch.empty((gb // 4) << 30, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    del keep
for _ in range(8):
    eng.train_step(images, b[1], b[2], b[3])
import gc, time
gcs = []
def _cb(phase, info, _t=[0.0]):
    if phase == 'start':
        _t[0] = time.perf_counter()
    else:
        gcs.append((info['generation'], round((time.perf_counter() - _t[0]) * 1e3, 1), cur[0]))
cur = [0]
gc.callbacks.append(_cb)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
mall = []
ev[0].record()
for i in range(n):
    cur[0] = i
    t0 = time.perf_counter()
    eng.train_step(images, b[1], b[2], b[3])
    host.append(round((time.perf_counter() - t0) * 1e3, 1))
    mall.append(torch.cuda.memory_stats().get('num_device_alloc', -1))
    ev[i + 1].record()
torch.cuda.synchronize()
gc.callbacks.remove(_cb)
print(json.dumps({'device_mallocs_per_step': [mall[i] - (mall[i - 1] if i else mall[0]) for i in range(n)], 'host_ms': host, 'gc_gen_ms_step': [g for g in gcs if g[0] >= 1 or g[1] > 2.0], 'n_gc': len(gcs), 'tracked_objects': len(gc.get_objects())}))
ms = [round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(n)]
st = torch.cuda.memory_stats()
slow = [i for i, t in enumerate(ms) if t > 1.04 * min(ms)]
print(json.dumps({'slow_steps': slow, 'n_slow': len(slow), 'median': sorted(ms)[n // 2], 'mean': round(sum(ms) / n, 3), 'min': min(ms), 'max': max(ms),
                  'alloc_retries': st.get('num_alloc_retries'), 'segments': st.get('segment.all.allocated'),
                  'reserved_GB': round(torch.cuda.memory_reserved() / 2**30, 2)}))
