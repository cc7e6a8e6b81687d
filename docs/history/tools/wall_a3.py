#!/usr/bin/env python3
"""Wall time of the client contrast step (A3 + A4, fwd + bwd) without any profiler: steady-state step time (device-bound
or host-bound), and the host-only cost of issuing one step (time until the Python call returns)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from creamfl_amd.algorithms.contrast import client_contrast_loss

def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device='cuda'), dim=-1)

for (B, M, D) in [(128, 50000, 256), (32, 50000, 256), (128, 50000, 768)]:
    G, Gs = unit(M, D), unit(M, D)
    idx = torch.randperm(M, device='cuda')[:B]
    f = unit(B, D).requires_grad_(True)
    fo = unit(B, D)
    def step():
        loss, _, _ = client_contrast_loss(f, Gs, G, idx, fo)
        loss.backward()
    for _ in range(10): step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / n
    rec = {'B': B, 'M': M, 'D': D, 'wall_us_per_step': round(t_wall * 1e6, 1), 'host_issue_us_per_step': round(t_issue * 1e6, 1)}
    # the same step captured in a HIP graph (the C-ABI launches allocate nothing and never synchronise): launch-bound inner
    # loops replay device-bound
    try:
        f.grad = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        f.grad = None
        with torch.cuda.graph(graph):
            step()
        for _ in range(5): graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): graph.replay()
        torch.cuda.synchronize()
        rec['graph_wall_us_per_step'] = round((time.perf_counter() - t0) / n * 1e6, 1)
    except Exception as e:                                    # noqa: BLE001
        rec['graph_error'] = repr(e)[:200]
    print(json.dumps(rec))
