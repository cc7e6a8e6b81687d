#!/usr/bin/env python3
"""BASELINE.json configs[4] at full encoder size on ONE GPU: ViT-B/16 + BERT-large PCME, d = 768, bf16 trunks, per-GPU batch
--batch (64 and 256 are kept in profiles/), with the model-FLOPs utilisation of the server step.
Times the server contrastive step (forward -> pair loss -> backward -> clip -> AdamP) and the client-style inter + intra step
against a 50 000-row bank (the D = 768 bank kernel).  One JSON line; run under `rocprofv3 --kernel-trace --stats` for the
per-kernel table kept in profiles/."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401  (library set-up: creamfl_amd/runtime.py)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    args = ap.parse_args()
    from creamfl_amd import _lib
    from creamfl_amd.algorithms.contrast import client_contrast_loss
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    _lib.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(41)
    cfg = default_config(embed_dim=768, cnn_type='vit_b_16', not_bert=False)
    cfg.model.bert_name = 'bert-large-uncased'
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(args.batch, dev, seed=42, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)

    def step():
        return eng.train_step(images, b[1], b[2], b[3])

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as _bench                                       # the FLOP counter of the bench line (module hooks + attention)
    fwd = _bench.forward_flops(eng, images, b[1], b[2], b[3])
    fwd = fwd['total'] if isinstance(fwd, dict) else fwd            # (by tower since round 5)
    vit = eng.model.img_enc.cnn                                   # + the ViT's QK^T / PV, which forward_flops does not see
    if hasattr(vit, 'layers'):
        Lp = (images.shape[2] // vit.patch) * (images.shape[3] // vit.patch)
        fwd += len(vit.layers) * 4 * args.batch * Lp * Lp * vit.out_dim
        fwd += 2 * args.batch * Lp * (vit.patch ** 2 * 3) * vit.out_dim          # the patch projection runs as F.linear (no module hook)
        fwd += 2 * args.batch * (Lp - 49) * vit.out_dim * (vit.out_dim // 2)      # PIE w_1 over Lp positions (the counter assumes 49)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # the client side of configs[4]: inter + intra contrast, weight 0.5, against a 50 000 x 768 bank
    gen = torch.Generator().manual_seed(43)
    G = torch.nn.functional.normalize(torch.randn(50000, 768, generator=gen), dim=-1).to(dev)
    Gs = torch.nn.functional.normalize(torch.randn(50000, 768, generator=gen), dim=-1).to(dev)
    idx = torch.randperm(50000, generator=gen)[:args.batch].to(dev)
    f = torch.nn.functional.normalize(torch.randn(args.batch, 768, generator=gen), dim=-1).to(dev).requires_grad_(True)
    fo = torch.nn.functional.normalize(torch.randn(args.batch, 768, generator=gen), dim=-1).to(dev)

    def cstep():
        l, _, _ = client_contrast_loss(f, Gs, G, idx, fo, interintra_weight=0.5)
        l.backward()

    for _ in range(5):
        cstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        cstep()
    torch.cuda.synchronize()
    cus = (time.perf_counter() - t0) / 50 * 1e6
    from creamfl_amd import ops as _ops
    builds = _ops.BANK_IMAGE_BUILDS[0]
    n_params = sum(p.numel() for p in eng.model.parameters())
    print(json.dumps({'config': 'BASELINE configs[4]: ViT-B/16 + BERT-large, d=768, batch %d, bf16 trunks' % args.batch,
                      'server_step_ms': round(ms, 2), 'pairs_per_s': round(args.batch / ms * 1e3, 1),
                      'loss': round(float(loss), 4), 'params_M': round(n_params / 1e6, 1),
                      'mfu': round(3.0 * fwd / (ms * 1e-3) / 2.5e15, 4), 'model_tflop_per_step': round(3.0 * fwd / 1e12, 2),
                      'client_contrast_step_us_wall': round(cus, 1), 'bank_image_builds': builds,
                      'vit_glue': 'aten' if os.environ.get('CFL_NO_VIT_FUSE') else 'fused (csrc/bertfuse.hip pre-LN chain)'}))


if __name__ == '__main__':
    main()
