"""Print the loss / gradient-norm trajectory of the bench step (diagnostic; `BF16W=0` keeps fp32 trunk weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
from creamfl_amd.utils.config import default_config
from creamfl_amd.utils.synthetic import coco_batch
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
B = int(os.environ.get('B', 256)); steps = int(os.environ.get('STEPS', 16))
torch.manual_seed(1234)
cfg = default_config(embed_dim=512, cnn_type=os.environ.get('CNN', 'resnet101'), not_bert=False)
eng = TrainerEngine(device=dev); eng.create(cfg, {'<pad>': 0}, None, False); eng.model_to_device()
eng.to_half(bf16_weights=os.environ.get('BF16W', '1') == '1')
eng.model.train()
b = coco_batch(B, dev, seed=1234, bert=True)
images = b[0].contiguous(memory_format=torch.channels_last)
for it in range(steps):
    loss, ld = eng.train_step(images, b[1], b[2], b[3])
    bad = [n for n, p in eng.model.named_parameters() if not torch.isfinite(p.float()).all()]
    gbad = [n for n, p in eng.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad.float()).all()]
    print('it%02d loss %.4f shift %.4f nscale %.4f bad_params %d bad_grads %d %s' % (
        it, float(loss), float(eng.criterion.shift), float(eng.criterion.negative_scale), len(bad), len(gbad),
        (bad[:3], gbad[:3])), flush=True)
