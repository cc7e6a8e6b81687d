"""Does torch's dense embedding backward synchronise the host?  A long matmul queue is put in front of it and the HOST time of
the call is measured: microseconds if it only enqueues, the queue's milliseconds if it waits for the device (the sort /
unique-by-key path above 3072 indices).  Same for ops.embedding_lookup's index_add_ backward."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from creamfl_amd import ops, runtime  # noqa: E402


def main():
    runtime.configure()
    dev = torch.device('cuda:0')
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    V, E = 30522, 768
    w = torch.randn(V, E, device=dev, requires_grad=True)
    emb = torch.nn.Embedding(V, E, padding_idx=0).to(dev)
    out = []
    for n in (2048, 3072, 3073, 4096, 8192):
        idx = torch.randint(0, V, (n,), device=dev)
        g = torch.randn(n, E, device=dev)
        for name, fn in (('torch', lambda: torch.ops.aten.embedding_dense_backward(g, idx, V, 0, False)),
                         ('index_add', lambda: torch.zeros(V, E, device=dev).index_add_(0, idx, g))):
            fn()
            torch.cuda.synchronize()
            for _ in range(12):
                b = a @ a                                  # ~1.1 TFLOP each: a queue of >= 10 ms
            t0 = time.perf_counter()
            fn()
            host_us = (time.perf_counter() - t0) * 1e6
            torch.cuda.synchronize()
            drain_ms = (time.perf_counter() - t0) * 1e3
            out.append({'indices': n, 'backward': name, 'host_us_of_the_call_behind_a_busy_queue': round(host_us, 1),
                        'queue_drained_after_ms': round(drain_ms, 2)})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
