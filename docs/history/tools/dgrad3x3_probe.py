#!/usr/bin/env python3
"""3x3 / stride 1 / pad 1 data gradient: MIOpen's backward-data kernel vs its FORWARD kernel on the rotated, transposed weight
(dX = conv2d(dY, W'), W'[ci, co, kh, kw] = W[co, ci, 2-kh, 2-kw]).  ResNet-101 shapes, NHWC bf16, batch 256."""
import os, sys, json
os.environ.setdefault('MIOPEN_FIND_MODE', os.environ.get('FM', '1'))
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


for (H, C) in [(56, 64), (28, 128), (14, 256), (7, 512)]:
    x = torch.randn(N, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    args = (dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    ref = torch.ops.aten.convolution_backward(*args, [True, False, False])[0]
    wr = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    alt = F.conv2d(dy, wr, None, 1, 1)
    err = float((alt.float() - ref.float()).abs().max() / ref.float().abs().max())
    t_ref = t_us(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False]))
    t_alt = t_us(lambda: F.conv2d(dy, wr, None, 1, 1))
    t_fwd = t_us(lambda: F.conv2d(x, w, None, 1, 1))
    t_wg = t_us(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
    print(json.dumps({'conv': f'3x3 {H}x{H} {C}->{C}', 'miopen_dgrad_us': round(t_ref, 1), 'fwd_kernel_as_dgrad_us': round(t_alt, 1),
                      'fwd_us': round(t_fwd, 1), 'wgrad_us': round(t_wg, 1), 'rel_err': round(err, 5)}), flush=True)
