#!/usr/bin/env python3
"""Timing probe of the fused bank kernels: fixed vs per-chunk cost (rocprof-free: HIP events of the library profiler
minus its measured floor).  GPU only."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from creamfl_amd import _lib, ops

def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device='cuda'), dim=-1)

def run(B, M, D, grad):
    G = unit(M, D); f = unit(B, D).requires_grad_(grad); idx = torch.randperm(M, device='cuda')[:B]
    def step():
        if grad:
            loss, lse, pos = ops.inter_contrast(f, G, idx)
            loss.backward()
        else:
            with torch.no_grad():
                ops.inter_contrast(f, G, idx)
    for _ in range(3): step()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): step()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    return {k: round(ms / n * 1e3, 1) for k, (n, ms) in _lib.prof_query().items()}

for grad in (False, True):
    for B in (128, 32):
        for M in (3200, 12800, 51200, 204800):
            print(json.dumps({'grad': grad, 'B': B, 'M': M, 'D': 256, 'us': run(B, M, 256, grad)}))
