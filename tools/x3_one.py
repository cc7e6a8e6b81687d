#!/usr/bin/env python3
"""One BasicBlock shape of the clients' ResNet-18 through the three fp32-accurate convolution kernels (csrc/conv3x3_x3.hip forward,
csrc/wgrad3x3_x3.hip) a few times: the command tools/pmc_sq.sh / tools/pmc_run.sh profile.   python tools/x3_one.py --hw 28 --c 128"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from creamfl_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--hw', type=int, default=28)
ap.add_argument('--c', type=int, default=128)
ap.add_argument('--n', type=int, default=128)
ap.add_argument('--variant', type=int, default=0)
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
cl = torch.channels_last
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(a.n, a.c, a.hw, a.hw, generator=g, device='cuda').contiguous(memory_format=cl)
w = (torch.randn(a.c, a.c, 3, 3, generator=g, device='cuda') / (3.0 * a.c ** 0.5)).contiguous(memory_format=cl)
dy = torch.randn(a.n, a.c, a.hw, a.hw, generator=g, device='cuda').contiguous(memory_format=cl)
for _ in range(a.iters):
    ops.conv3x3_x3_forward(x, w, a.variant)
    ops.conv3x3_x3_wgrad(dy, x, w)
torch.cuda.synchronize()
