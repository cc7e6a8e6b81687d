import sys, os, json, torch
sys.path.insert(0, '/root/repo')
from creamfl_amd import _lib, ops
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device='cuda'), dim=-1)
def run(B, M, D):
    G = unit(M, D); f = unit(B, D).requires_grad_(True); idx = torch.randperm(M, device='cuda')[:B]
    def step():
        loss, lse, pos = ops.inter_contrast(f, G, idx); loss.backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(20): step()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    return {k: round(ms / n * 1e3, 1) for k, (n, ms) in _lib.prof_query().items()}
for (B, M, D) in [(128, 51200, 256), (128, 204800, 256), (128, 12800, 256)]:
    print(json.dumps({'B': B, 'M': M, 'D': D, 'us': run(B, M, D)}))
