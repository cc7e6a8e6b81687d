"""Where do the 44 ms per public batch of the server's global-training phase go INSIDE a federation process (bench.py --config 2:
35-45 ms) when a process holding only the server engine takes 27-28 ms (docs/history/tools/server_graph_ab.py)?  Per-batch host timestamps of
TrainerEngine.train_step inside MMFL.train (no synchronisation added; the phase's wall time by one synchronisation at its end),
for two rounds of the bench's federation."""
import argparse
import json
import os
import random
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, default=50)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--sync-every', type=int, default=0, help='> 0: synchronise after every n-th step (GPU time per step, perturbs the loop)')
    ap.add_argument('--mini-first', type=int, default=0, help='the miniature warm-up federation of the bench first')
    ap.add_argument('--measure-first', type=int, default=0, help='the three client step measurements of the bench first')
    ap.add_argument('--empty-cache', type=int, default=0, help='torch.cuda.empty_cache() before every server phase')
    ap.add_argument('--gc-off', type=int, default=0, help='gc.disable() while a server phase runs')
    ap.add_argument('--tag', default='')
    a = ap.parse_args()
    import bench_clients
    p2 = argparse.ArgumentParser()
    bench_clients.add_arguments(p2)
    fa = p2.parse_args([])
    fa.steps, fa.warmup = 5, 2
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    algo, _ = bench_clients.build_federation(fa, dev, 128 * a.batches)
    if a.measure_first:
        from creamfl_amd.utils.synthetic import coco_batch_on_device
        g = torch.Generator(device=dev).manual_seed(4321)
        unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g, device=dev), dim=-1)
        banks = (unit(50000, 256), unit(50000, 256))
        batch = coco_batch_on_device(128, dev, seed=1234, img=224)
        for kind in ('img', 'txt', 'mm'):
            bench_clients.measure_client(bench_clients.first_of_kind(algo, kind), kind, banks, batch, dev, 10, 3, False)
    if a.mini_first:
        mini, _ = bench_clients.build_federation(fa, dev, 4 * 128, mini=True)
        random.seed(4321)
        mini.train(0)
        torch.cuda.synchronize()
        del mini
        torch.cuda.empty_cache()
    eng = algo.engine
    stamps, phases = [], []
    orig_step, orig_train = eng.train_step, eng.train

    def step(*args, **kw):
        out = orig_step(*args, **kw)
        if a.sync_every and len(stamps) % a.sync_every == a.sync_every - 1:
            torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        return out

    def train(*args, **kw):
        import gc
        del stamps[:]
        torch.cuda.synchronize()
        if a.empty_cache:
            torch.cuda.empty_cache()
        if a.gc_off:
            gc.disable()
        t0 = time.perf_counter()
        stamps.append(t0)
        orig_train(*args, **kw)
        t_issue = time.perf_counter()
        if a.gc_off:
            gc.enable()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d = [round((y - x) * 1e3, 2) for x, y in zip(stamps, stamps[1:])]
        phases.append({'batches': len(d), 'wall_ms_per_batch': round((t1 - t0) / max(1, len(d)) * 1e3, 2),
                       'issue_ms_per_batch': round((t_issue - t0) / max(1, len(d)) * 1e3, 2), 'drain_ms': round((t1 - t_issue) * 1e3, 1),
                       'first6_ms': d[:6], 'median_rest_ms': round(statistics.median(d[6:]), 2) if len(d) > 7 else None,
                       'p90_rest_ms': round(sorted(d[6:])[int(0.9 * len(d[6:]))], 2) if len(d) > 7 else None,
                       'max_rest_ms': max(d[6:]) if len(d) > 7 else None,
                       'mem_gb': {'allocated': round(torch.cuda.memory_allocated() / 2 ** 30, 2),
                                  'reserved': round(torch.cuda.memory_reserved() / 2 ** 30, 2)}})
    eng.train_step, eng.train = step, train
    for r in range(a.rounds):
        random.seed(1234)
        algo.train(r)
    print(json.dumps({'tag': a.tag, 'empty_cache': a.empty_cache, 'gc_off': a.gc_off, 'inline_loader': os.environ.get('CFL_PREFETCH_INLINE'), 'alloc_conf': os.environ.get('PYTORCH_HIP_ALLOC_CONF') or os.environ.get('PYTORCH_CUDA_ALLOC_CONF'), 'mini_first': a.mini_first, 'measure_first': a.measure_first, 'threads': __import__('threading').active_count(), 'sync_every': a.sync_every, 'phases': phases}), flush=True)


if __name__ == '__main__':
    main()
