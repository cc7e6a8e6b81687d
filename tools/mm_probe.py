#!/usr/bin/env python3
"""1x1 convolutions as plain GEMMs on the NHWC-flattened activation: hipBLASLt (torch.mm) vs MIOpen conv, bf16."""
import os, sys, json
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda'
N = 256


def t_us(fn, it=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for (H, Ci, Co) in [(56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128), (28, 512, 256),
                    (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)]:
    R = N * H * H
    x = torch.randn(R, Ci, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(Co, Ci, device=dev) * 0.05).to(torch.bfloat16).requires_grad_(True)
    y = x @ w.t()
    dy = torch.randn_like(y)
    fwd = t_us(lambda: x @ w.t())
    dgrad = t_us(lambda: dy @ w)
    wgrad = t_us(lambda: dy.t() @ x)
    x4 = x.detach().view(N, H, H, Ci).permute(0, 3, 1, 2).requires_grad_(True)
    w4 = w.detach().view(Co, Ci, 1, 1).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y4 = F.conv2d(x4, w4)
    dy4 = dy.view(N, H, H, Co).permute(0, 3, 1, 2)
    cf = t_us(lambda: F.conv2d(x4, w4))
    cb = t_us(lambda: torch.autograd.grad(y4, (x4, w4), dy4, retain_graph=True))
    print(json.dumps({'shape': f'{H}x{H} {Ci}->{Co}', 'mm_fwd': round(fwd, 1), 'conv_fwd': round(cf, 1), 'mm_dgrad': round(dgrad, 1),
                      'mm_wgrad': round(wgrad, 1), 'mm_bwd': round(dgrad + wgrad, 1), 'conv_bwd': round(cb, 1),
                      'roof_fwd_us': round((R * (Ci + Co) * 2) / 6.0e6, 1)}), flush=True)
