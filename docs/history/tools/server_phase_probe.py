"""The server's global-training phase of a configs[2] round as the round runs it -- TrainerEngine.train over a device-born public
loader at the reference's public batch of 128 -- against the resident step of tools/host_profile_step.py: ms per batch, and where
the host time of the loop goes (cProfile; the backward runs on the calling thread)."""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--batches', type=int, default=50)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--federation', type=int, default=0, help='1: the engine of a whole configs[2] federation (25 clients built beside it)')
    a = ap.parse_args()
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import DeviceCocoLoader
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    if a.federation:
        import argparse as _ap
        import bench_clients
        p2 = _ap.ArgumentParser()
        bench_clients.add_arguments(p2)
        fa = p2.parse_args([])
        fa.steps, fa.warmup = 5, 2
        algo, _ = bench_clients.build_federation(fa, dev, a.batch * a.batches)
        eng = algo.engine
        loader = algo._dataloaders[algo._pub_key(False)]
    else:
        cfg = default_config(embed_dim=256, cnn_type='resnet101', not_bert=False)
        eng = TrainerEngine(device=dev)
        eng.create(cfg, {'<pad>': 0}, None, False)
        eng.model_to_device()
        eng.to_half()
        loader = DeviceCocoLoader(a.batch * a.batches, a.batch, seed=1, device=dev)

    def phase():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train(loader)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.batches * 1e3

    out = {'batch': a.batch, 'batches': a.batches, 'federation': a.federation, 'first_pass_ms_per_batch': round(phase(), 2)}
    out['second_pass_ms_per_batch'] = round(phase(), 2)
    out['third_pass_ms_per_batch'] = round(phase(), 2)
    if a.federation:
        import random
        random.seed(1234)
        algo.train(0)
        out['after_a_round_ms_per_batch'] = [round(phase(), 2), round(phase(), 2)]
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out['after_empty_cache_ms_per_batch'] = [round(phase(), 2), round(phase(), 2)]
        from creamfl_amd import _lib
        out['profiler_enabled'] = bool(_lib.load().cfl_prof_enabled()) if hasattr(_lib.load(), 'cfl_prof_enabled') else None
        out['threads'] = __import__('threading').active_count()
        out['mem_gb'] = {'allocated': round(torch.cuda.memory_allocated() / 2**30, 2), 'reserved': round(torch.cuda.memory_reserved() / 2**30, 2)}
    # the same step on ONE resident batch
    b = next(iter(loader))
    eng.model.train()
    for _ in range(5):
        eng.train_step(b[0], b[1], b[2], b[3], gather=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.batches):
        eng.train_step(b[0], b[1], b[2], b[3], gather=False)
    torch.cuda.synchronize()
    out['resident_step_ms'] = round((time.perf_counter() - t0) / a.batches * 1e3, 2)
    # loader alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in loader:
        pass
    torch.cuda.synchronize()
    out['loader_alone_ms_per_batch'] = round((time.perf_counter() - t0) / a.batches * 1e3, 2)
    print(json.dumps(out))
    pr = cProfile.Profile()
    pr.enable()
    phase()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(a.top)
    print('\n'.join(line[:170] for line in s.getvalue().splitlines()))


if __name__ == '__main__':
    main()
