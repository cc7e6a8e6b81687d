"""Where does the host stall?  A sampler thread records the main thread's Python stack whenever the current training step has
been on the host for more than 60 ms (a step normally takes ~35 ms of host time)."""
import os, sys, json, time, threading, traceback, collections
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
for _ in range(8):
    eng.train_step(images, b[1], b[2], b[3])
main_id = threading.main_thread().ident
state = {'t0': None, 'run': True}
hits = collections.Counter()
def sampler():
    while state['run']:
        time.sleep(0.004)
        t0 = state['t0']
        if t0 is not None and time.perf_counter() - t0 > 0.060:
            fr = sys._current_frames().get(main_id)
            if fr is not None:
                st = traceback.extract_stack(fr)
                key = ' <- '.join('%s:%d:%s' % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(st[-6:]))
                hits[key] += 1
th = threading.Thread(target=sampler, daemon=True)
th.start()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
host = []
for i in range(n):
    state['t0'] = time.perf_counter()
    eng.train_step(images, b[1], b[2], b[3])
    host.append(round((time.perf_counter() - state['t0']) * 1e3, 1))
    state['t0'] = None
state['run'] = False
torch.cuda.synchronize()
print(json.dumps({'host_ms': host}))
for k, v in hits.most_common(12):
    print(v, k)
