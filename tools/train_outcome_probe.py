"""Training-outcome probe (VERDICT r4 missing #4): the SAME PCME trained on a learnable synthetic retrieval task
(tests/learnable_task.py) with bf16 fused trunks (the bench's code path) and with fp32 trunks, from the same initial state; then
COCOEvaluator.evaluate on held-out samples.  Prints one JSON line per run.

    python tools/train_outcome_probe.py [--steps 400] [--batch 32] [--n-id 200] [--lr 2e-4] [--noise 0.3]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from creamfl_amd import runtime  # noqa: E402

runtime.configure_env()
import torch  # noqa: E402


def train_and_eval(task, steps, batch, lr, fp32, state, dev, cnn='resnet18', dim=64, n_eval=200, log_every=50, seed=3):
    from creamfl_amd.algorithms.eval_coco import COCOEvaluator
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from learnable_task import EvalLoader
    torch.manual_seed(seed)
    cfg = default_config(embed_dim=dim, cnn_type=cnn, not_bert=False)
    cfg.model.bert_name = 'bert-mini'
    cfg.optimizer.learning_rate = lr
    ev = COCOEvaluator(eval_method='matmul', verbose=False, eval_device=str(dev), extract_device=str(dev), n_crossfolds=5)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, ev, False)
    for m in eng.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                      # (the two paths draw their dropout masks differently: not what is compared here)
    if state is not None:
        eng.model.load_state_dict(state)
    state0 = copy.deepcopy(eng.model.state_dict())
    eng.model_to_device()
    if not fp32:
        eng.to_half()
    eng.model.train()
    losses = []
    t0 = time.time()
    for s in range(steps):
        b = task.train_batch(s, batch)
        images = b[0] if fp32 else b[0].contiguous(memory_format=torch.channels_last)
        loss, _ = eng.train_step(images, b[1], b[2], b[3])
        if s % log_every == 0 or s == steps - 1:
            losses.append(round(float(loss.detach()), 3))
    torch.cuda.synchronize()
    dt = time.time() - t0
    scores = eng.evaluate({'te': EvalLoader(task, n_eval=n_eval)}, n_crossfolds=5, n_images_per_crossfold=n_eval // 5,
                          n_captions_per_crossfold=n_eval)['te']
    return {'fp32': fp32, 'i2t_r1': scores['i2t']['recall_1'], 't2i_r1': scores['t2i']['recall_1'], 'i2t_r5': scores['i2t']['recall_5'],
            'fold_i2t_r1': scores['n_fold']['i2t']['recall_1'], 'fold_t2i_r1': scores['n_fold']['t2i']['recall_1'],
            'losses': losses, 'train_s': round(dt, 1)}, state0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--n-id', type=int, default=200)
    ap.add_argument('--lr', type=float, default=2e-4)
    ap.add_argument('--noise', type=float, default=0.3)
    ap.add_argument('--img', type=int, default=64)
    ap.add_argument('--cnn', default='resnet18')
    ap.add_argument('--seeds', type=int, default=1, help='model-initialisation seeds 3, 4, ...: one bf16 + one fp32 run per seed')
    ap.add_argument('--n-eval', type=int, default=0, help='held-out identities evaluated (0 = all; a multiple of 5)')
    ap.add_argument('--caption-swap', type=float, default=0.0, help='share of captions carrying another identity\'s signature')
    args = ap.parse_args()
    from learnable_task import LearnableTask
    dev = torch.device('cuda', 0)
    task = LearnableTask(n_id=args.n_id, img=args.img, seed=0, noise=args.noise, device=dev, caption_swap=args.caption_swap)
    n_eval = args.n_eval or args.n_id
    for seed in range(3, 3 + args.seeds):
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            a, state = train_and_eval(task, args.steps, args.batch, args.lr, False, None, dev, cnn=args.cnn, n_eval=n_eval, seed=seed)
            print(json.dumps(dict(a, seed=seed, **vars(args))), flush=True)
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):       # fp32 immediate mode = fallback kernels (~0.4 s per step)
            b, _ = train_and_eval(task, args.steps, args.batch, args.lr, True, state, dev, cnn=args.cnn, n_eval=n_eval, seed=seed)
            print(json.dumps(dict(b, seed=seed, **vars(args))), flush=True)


if __name__ == '__main__':
    main()
