#!/usr/bin/env python3
"""How far are MIOpen's convolutions from the HBM roofline on ResNet-101's shapes (NHWC bf16, batch 256)?
Prints time and effective GB/s (algorithmic bytes: read in + write out for fwd; 2 reads + 1 write for each of
dgrad / wgrad).   python tools/conv_probe.py"""
import os, sys, json
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda'
N = int(os.environ.get('N', 256))


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


shapes = [  # (H_in, Ci, Co, k, stride)
    (56, 64, 64, 1, 1), (56, 64, 64, 3, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1),
    (56, 256, 128, 1, 1), (56, 128, 128, 3, 2), (28, 128, 512, 1, 1), (56, 256, 512, 1, 2), (28, 512, 128, 1, 1), (28, 128, 128, 3, 1),
    (28, 512, 256, 1, 1), (28, 256, 256, 3, 2), (14, 256, 1024, 1, 1), (28, 512, 1024, 1, 2), (14, 1024, 256, 1, 1), (14, 256, 256, 3, 1),
    (14, 1024, 512, 1, 1), (14, 512, 512, 3, 2), (7, 512, 2048, 1, 1), (14, 1024, 2048, 1, 2), (7, 2048, 512, 1, 1), (7, 512, 512, 3, 1),
]
if os.environ.get('ONLY_LAYER3'):
    shapes = [sh for sh in shapes if sh[0] == 14 and sh[4] == 1]
for (H, Ci, Co, k, s) in shapes:
    x = torch.randn(N, Ci, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    y = F.conv2d(x, w, None, s, k // 2)
    dy = torch.randn_like(y)
    fwd = t_us(lambda: F.conv2d(x, w, None, s, k // 2))

    def bwd():
        torch.autograd.grad(y, (x, w), dy, retain_graph=True)
    b = t_us(bwd)
    bin_, bout = N * H * H * Ci * 2, N * Ho * Ho * Co * 2
    print(json.dumps({'conv': f'{k}x{k}s{s} {H}x{H} {Ci}->{Co}', 'fwd_us': round(fwd, 1), 'bwd_us': round(b, 1),
                      'fwd_GBps': round((bin_ + bout) / fwd / 1e3), 'bwd_GBps': round((2 * bin_ + 2 * bout + bin_) / b / 1e3),
                      'fwd_TFLOPs': round(2 * N * Ho * Ho * Co * Ci * k * k / fwd / 1e6, 1)}), flush=True)
