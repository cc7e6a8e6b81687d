"""Where does the HOST time of one client contrast step (A3 + A4, fwd + bwd) go?  cProfile over N eager steps (device work is
queued asynchronously; the profile is the Python / ctypes / allocator side only) + the plain wall numbers of tools/wall_a3.py.
    python tools/host_prof_a3.py [--b 128] [--d 256] [--n 2000]"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401
import torch  # noqa: E402
from creamfl_amd.algorithms.contrast import client_contrast_loss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--b', type=int, default=128)
ap.add_argument('--m', type=int, default=50000)
ap.add_argument('--d', type=int, default=256)
ap.add_argument('--n', type=int, default=2000)
ap.add_argument('--top', type=int, default=28)
args = ap.parse_args()


def unit(*s):
    return torch.nn.functional.normalize(torch.randn(*s, device='cuda'), dim=-1)


G, Gs = unit(args.m, args.d), unit(args.m, args.d)
idx = torch.randperm(args.m, device='cuda')[:args.b]
f = unit(args.b, args.d).requires_grad_(True)
fo = unit(args.b, args.d)


def step():
    loss, _, _ = client_contrast_loss(f, Gs, G, idx, fo)
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.n):
    step()
issue = (time.perf_counter() - t0) / args.n
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.n
print(json.dumps({'B': args.b, 'M': args.m, 'D': args.d, 'wall_us': round(wall * 1e6, 1), 'host_issue_us': round(issue * 1e6, 1)}))
pr = cProfile.Profile()
pr.enable()
for _ in range(args.n):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(args.top)
print('\n'.join(ln[:150] for ln in s.getvalue().splitlines()[:args.top + 12]))
