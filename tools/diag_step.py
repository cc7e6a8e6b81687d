import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def log(m):
    print(f'[{time.time()-t0:7.1f}s] {m}', flush=True)
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch
dev = torch.device('cuda:0')
if os.environ.get('BENCHMARK'): torch.backends.cudnn.benchmark = True
log(f'cpu_count {os.cpu_count()} affinity {len(os.sched_getaffinity(0))} threads {torch.get_num_threads()}')
B = int(os.environ.get('B', 256)); cnn = os.environ.get('CNN', 'resnet101'); dt = os.environ.get('DT', 'bf16')
cfg = default_config(embed_dim=512, cnn_type=cnn)
eng = TrainerEngine(device=dev); eng.create(cfg, {'<pad>': 0}, None, False); eng.model_to_device()
if dt == 'bf16': eng.to_half()
if os.environ.get('NATIVE_BN'):
    import torch.nn.functional as F
    class NBN(torch.nn.BatchNorm2d):
        def forward(self, x):
            with torch.backends.cudnn.flags(enabled=False):
                return super().forward(x)
    def swap(m):
        for n, c in m.named_children():
            if isinstance(c, torch.nn.BatchNorm2d):
                nb = NBN(c.num_features).to(dev); nb.load_state_dict(c.state_dict()); setattr(m, n, nb)
            else: swap(c)
    swap(eng.model)
eng.model.train(); log('model ready')
b = coco_batch(B, dev, 1234, True); log('batch ready')
images = b[0].contiguous(memory_format=torch.channels_last) if dt == 'bf16' else b[0]
for it in range(6):
    torch.cuda.synchronize(); t = time.time()
    loss, _ = eng.forward_loss(images, b[1], None, b[3]); torch.cuda.synchronize(); t1 = time.time()
    eng.optimizer.zero_grad(set_to_none=True); loss.backward(); torch.cuda.synchronize(); t2 = time.time()
    torch.nn.utils.clip_grad_norm_(eng.model.parameters(), 2.0); torch.cuda.synchronize(); t3 = time.time()
    eng.optimizer.step(); torch.cuda.synchronize(); t4 = time.time()
    log(f'it{it}: fwd {t1-t:.3f} bwd {t2-t1:.3f} clip {t3-t2:.3f} opt {t4-t3:.3f} loss {loss.item():.3f}')
