#!/usr/bin/env python3
"""ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 channels, 224x224, batch 256, NHWC bf16): MIOpen on the problem as written vs the
same convolution after space-to-depth (4x4 / stride 1 on [112, 112, 12 (+4 zero)] -- 7 = 2 * 4 - 1 taps per axis)."""
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


def s2d_input(x, cpad):
    """[N, 3, 224, 224] channels_last -> [N, 12 + cpad, 112, 112] channels_last; channel = (ph * 2 + pw) * 3 + c."""
    n = x.shape[0]
    v = x.permute(0, 2, 3, 1).reshape(n, 112, 2, 112, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(n, 112, 112, 12)
    if cpad:
        v = F.pad(v, (0, cpad))
    return v.permute(0, 3, 1, 2)


def s2d_weight(w, cpad):
    """[64, 3, 7, 7] -> [64, 12 + cpad, 4, 4]: tap kh = 2 a + p + 3 with a in -2..1, p in 0..1 (kh = -1 does not exist: zero)."""
    co = w.shape[0]
    wp = F.pad(w, (1, 0, 1, 0))                               # taps -1..6 -> index 0..7 ; index = kh + 1 = 2 (a + 2) + p
    v = wp.reshape(co, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(co, 12, 4, 4)     # [co, (ph, pw, c), a, b]
    if cpad:
        v = F.pad(v, (0, 0, 0, 0, 0, cpad))
    return v.contiguous(memory_format=torch.channels_last)


x = torch.randn(N, 3, 224, 224, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ref = F.conv2d(x, w, None, 2, 3)
dy = torch.randn_like(ref)
out = {'ref_fwd_us': round(t_us(lambda: F.conv2d(x, w, None, 2, 3)), 1)}
args = (dy, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1)
out['ref_wgrad_us'] = round(t_us(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False])), 1)
for cpad in (0, 4):
    x4 = s2d_input(x, cpad).contiguous(memory_format=torch.channels_last)
    w4 = s2d_weight(w, cpad)
    xp = F.pad(x4, (2, 1, 2, 1)).contiguous(memory_format=torch.channels_last)       # a = -2..1: two before, one after
    y = F.conv2d(xp, w4, None, 1, 0)
    err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
    tag = 'c%d' % (12 + cpad)
    out[tag + '_rel_err'] = round(err, 5)
    out[tag + '_s2d_us'] = round(t_us(lambda: F.pad(s2d_input(x, cpad), (2, 1, 2, 1)).contiguous(memory_format=torch.channels_last)), 1)
    out[tag + '_fwd_us'] = round(t_us(lambda: F.conv2d(xp, w4, None, 1, 0)), 1)
    a4 = (dy, xp, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
    out[tag + '_wgrad_us'] = round(t_us(lambda: torch.ops.aten.convolution_backward(*a4, [False, True, False])), 1)
print(json.dumps(out))
