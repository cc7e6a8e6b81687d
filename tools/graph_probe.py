"""Experiment: capture the whole server step (fwd + loss + bwd + clip + AdamP) in a HIP graph and replay it."""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = default_config(embed_dim=512, cnn_type='resnet101')
eng = TrainerEngine(device=dev); eng.create(cfg, {'<pad>': 0}, None, False); eng.model_to_device(); eng.to_half(); eng.model.train()
b = coco_batch(256, dev, seed=1234)
images = b[0].contiguous(memory_format=torch.channels_last)
def step():
    return eng.train_step(images, b[1], None, b[3])[0]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print('eager ms/step', (time.perf_counter() - t0) * 100)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print('graph ms/step', (time.perf_counter() - t0) * 100, 'loss', float(loss))
except Exception as e:
    import traceback; traceback.print_exc()
