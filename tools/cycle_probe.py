"""Which objects of one training step are only reclaimed by the cyclic garbage collector (reference cycles keep the step's
tensors alive past the step: memory creeps until a collection, the allocator has to grow, steps stall)."""
import os, sys, json, gc, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from creamfl_amd import _lib
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch
_lib.load()
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
cnn = sys.argv[1] if len(sys.argv) > 1 else 'resnet50'
cfg = default_config(embed_dim=512, cnn_type=cnn, not_bert=False)
eng = TrainerEngine(device=dev)
eng.create(cfg, {'<pad>': 0}, None, False)
eng.model_to_device(); eng.to_half(); eng.model.train()
b = coco_batch(32, dev, seed=1234, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)
for _ in range(3):
    eng.train_step(images, b[1], b[2], b[3])
torch.cuda.synchronize()
gc.collect()
gc.disable()
m0 = torch.cuda.memory_allocated()
eng.train_step(images, b[1], b[2], b[3])
torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
hist = collections.Counter(type(o).__module__ + '.' + type(o).__name__ for o in gc.garbage)
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
tb = sum(t.numel() * t.element_size() for t in tens if t.is_cuda)
print(json.dumps({'unreachable': n, 'allocated_growth_MB': round((m1 - m0) / 2**20, 1), 'garbage_tensors': len(tens),
                  'garbage_tensor_MB': round(tb / 2**20, 1), 'types': hist.most_common(25)}))
# who refers to the function contexts?
for o in gc.garbage:
    name = type(o).__name__
    if 'Backward' in name or 'Fn' in name:
        refs = [type(r).__name__ for r in gc.get_referrers(o) if r is not gc.garbage][:6]
        print(name, '<-', refs)
        break
