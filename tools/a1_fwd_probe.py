"""How much of the A1 forward at N = 4096, D = 512 is the coefficient-image stores?  The same forward with and without
`requires_grad` (without: no coefficient images are written), kernel times from the library profiler."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401
import torch  # noqa: E402
from creamfl_amd import _lib, ops  # noqa: E402

N, D = 4096, 512
g = torch.Generator(device='cuda').manual_seed(1)
I = torch.nn.functional.normalize(torch.randn(N, D, device='cuda', generator=g), dim=-1)
T = torch.nn.functional.normalize(I + 0.5 * torch.nn.functional.normalize(torch.randn(N, D, device='cuda', generator=g), dim=-1), dim=-1)
a = torch.tensor([15.0], device='cuda')
b = torch.tensor([15.0], device='cuda')
out = {}
for grad in (False, True):
    Ig = I.clone().requires_grad_(grad)
    for _ in range(3):
        ops.pair_loss(Ig, T, a, b)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(20):
        ops.pair_loss(Ig, T, a, b)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    out['grad' if grad else 'nograd'] = {k: round(ms / n * 1e3, 2) for k, (n, ms) in _lib.prof_query().items()}
print(json.dumps(out))
