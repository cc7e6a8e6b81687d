"""Which layout / precision would the image client's encoder want?  resnet18_client (D = 256), B = 128, 224 x 224: forward +
old-model forward (no grad) + backward + SGD per step, in the reference's form (fp32 NCHW), in fp32 channels_last, and under bf16
autocast + channels_last (below the reference's client precision: an opt-in at most).  One JSON line per mode.
    python tools/client_layout_probe.py [--steps 20]"""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from creamfl_amd import runtime  # noqa: E402

runtime.configure_env()
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--batch', type=int, default=128)
args = ap.parse_args()
runtime.configure()
from creamfl_amd.networks.resnet_client import resnet18_client  # noqa: E402
dev = torch.device('cuda', 0)
for mode in ('fp32 NCHW (reference form)', 'fp32 channels_last', 'bf16 autocast + channels_last'):
    torch.manual_seed(0)
    model = resnet18_client(pretrained=False, num_class=100, is_train=True, scale=128, mlp_local=False, embed_dim=256).to(dev).train()
    cl = 'channels_last' in mode
    if cl:
        model = model.to(memory_format=torch.channels_last)
    old = copy.deepcopy(model).eval()
    for m in (model, old):
        m.phase, m.is_train = 'extract_conv_feature', False
    opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-5)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    tgt = torch.nn.functional.normalize(torch.randn(args.batch, 256, device=dev), dim=-1)
    bf16 = mode.startswith('bf16')

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            f = model(x)
            with torch.no_grad():
                fo = old(x)
        loss = ((f.float() - tgt) ** 2).sum() + (f.float() * fo.float()).sum()
        loss.backward()
        opt.step()
        return loss
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({'mode': mode, 'ms_per_step': round(ms, 2), 'pairs_per_s': round(args.batch / ms * 1e3), 'warmup_s': round(warm, 1),
                      'loss': round(float(loss), 3)}), flush=True)
