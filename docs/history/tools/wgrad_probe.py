#!/usr/bin/env python3
"""MIOpen's data-gradient and weight-gradient times for the 1x1 convolutions of ResNet-101 (NHWC bf16, batch 256)."""
import os, json
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
import torch
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


for (H, Ci, Co) in [(56, 64, 64), (56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024), (14, 1024, 256),
                    (7, 512, 2048), (7, 2048, 512)]:
    x = torch.randn(N, Ci, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Co, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    args = (dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
    dg = t_us(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False]))
    wg = t_us(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
    M = N * H * H
    print(json.dumps({'shape': f'{H}x{H} {Ci}->{Co}', 'dgrad_us': round(dg, 1), 'wgrad_us': round(wg, 1),
                      'wgrad_roof_us': round(M * (Ci + Co) * 2 / 6.0e6, 1)}), flush=True)
