"""Which part of the multi-modal client's contrast step does not survive a HIP-graph capture?  Runs the step of
MMClientTrainer.contrast_step_fn through graphs.GraphedStep with parts switched off; one variant per process (a fault inside
hipStreamEndCapture takes the process down).  `python tools/mm_graph_probe.py` runs every variant as a child and prints one line each."""
import argparse
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(a):
    import copy
    from types import SimpleNamespace
    import torch
    from creamfl_amd.algorithms.MMClientTrainer import MMClientTrainer
    from creamfl_amd.algorithms.ClientTrainer import pad_captions
    from creamfl_amd.graphs import GraphedStep
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    dev = torch.device('cuda:0')
    M, D, bs = 136, 64, 16
    torch.manual_seed(11)
    args = SimpleNamespace(feature_dim=D, mlp_local=False, local_epochs=1, contrast_local_intra=True, contrast_local_inter=True,
                           interintra_weight=0.5, loss_scale=False, save_client=False, client_graph=1, mm_client_graph=1,
                           client_channels_last=0 if 'nchw' in a.off else 1)
    cfg = default_config(embed_dim=D, cnn_type='resnet18', not_bert=True)
    cfg.train.use_fp16 = False
    if 'clip' in a.off:
        cfg.train.grad_clip = 0
    t = MMClientTrainer(args, cfg, None, client=0, device=str(dev))
    t._to_device()
    t.model.train()
    t.old_model = copy.deepcopy(t.model).eval()
    if 'adamp' in a.off:
        t.optimizer = torch.optim.SGD(t.model.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    g_img = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    g_txt = torch.nn.functional.normalize(torch.randn(M, D, generator=gen), dim=-1).to(dev)
    b = coco_batch(bs, dev, seed=7, bert=False, img=64)
    images, captions, lens = b[0].to(dev), pad_captions(b[1].to(dev), 32), b[3].to(dev).to(torch.int64)
    d_idx = torch.randperm(M, generator=gen)[:bs].to(dev)
    step = t.contrast_step_fn(g_img, g_txt, 'intra' not in a.off, 'inter' not in a.off)
    if 'local' not in a.off:                          # a local PCME step first, as train_epoch does
        import contextlib
        side = torch.cuda.Stream(dev) if 'localside' in a.off else None
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            out = t._forward(t.model, images, captions, None, lens)
            if 'localnocrit' in a.off:
                loss = out['image_features'].square().sum() + out['caption_features'].square().sum()
            else:
                loss, _ = t.criterion(**out)
            if 'localnostep' in a.off:
                t.optimizer.zero_grad()
                loss.backward()
            else:
                t._step(loss)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
    if 'sync' in a.off:
        torch.cuda.synchronize()
    if 'emptycache' in a.off:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if 'gradnone' in a.off:
        t.optimizer.zero_grad(set_to_none=True)
    fn = lambda im, cap, ln, di: step(im, cap, None, ln, di)
    if 'imgonly' in a.off or 'txtonly' in a.off:
        # one tower only: the uni-modal loss on that tower's features
        from creamfl_amd.algorithms.contrast import client_contrast_loss
        key = 'image_features' if 'imgonly' in a.off else 'caption_features'

        def fn(im, cap, ln, di):
            t.optimizer.zero_grad()
            if 'imgonly' in a.off:
                f = t.model.img_enc(im.contiguous(memory_format=torch.channels_last))['embedding']
            else:
                f = t.model.txt_enc(cap, ln)['embedding']
            loss, _, _ = client_contrast_loss(f, g_img, g_txt, di, None, use_inter=True, use_intra=False)
            t._step(loss)
            return loss.detach()
    msgs = []
    gs = GraphedStep(fn, warmup=3, log=msgs.append, optimizer=t.optimizer if 'adamp' not in a.off else None)
    for _ in range(7):
        loss = gs(images, captions, lens, d_idx, device=dev)
    torch.cuda.synchronize()
    print(json.dumps({'off': a.off, 'replays': gs.replays, 'failed': gs.failed, 'loss': float(loss)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--off', default='')
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--variants', default='sync,emptycache,gradnone,localside,localnocrit,localnostep,imgonly+adamp,imgonly+adamp+clip')
    a = ap.parse_args()
    if a.child:
        a.off = [x for x in a.off.split('+') if x]
        child(a)
        return
    for v in a.variants.split(','):
        env = dict(os.environ)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--off', v], capture_output=True, text=True, env=env,
                           timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        print(json.dumps({'variant': v or 'full', 'rc': r.returncode, 'result': json.loads(line[-1]) if line else None,
                          'err_tail': None if r.returncode == 0 else r.stderr[-300:]}), flush=True)


if __name__ == '__main__':
    main()
