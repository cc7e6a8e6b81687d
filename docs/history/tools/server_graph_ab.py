"""--server_graph A/B of the server's global-training phase (TrainerEngine.train over a device-born public loader, the BASELINE
configs[1] model: ResNet-101 + BERT-base, d = 512, bf16 trunks) at the public batch of a configs[2] round (128) and at the
headline batch (256): ms per batch eager, ms per batch from the HIP graph (slope between a 20- and a 60-batch phase: the capture's
fixed cost -- three eager warm-up steps + the capture -- is reported separately), in ONE process per batch size."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--dim', type=int, default=512)
    ap.add_argument('--passes', type=int, default=2)
    ap.add_argument('--tag', default='')
    a = ap.parse_args()
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import DeviceCocoLoader
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=a.dim, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    msgs = []
    from types import SimpleNamespace
    eng.set_logger(SimpleNamespace(log=msgs.append, update_tracker=lambda *x, **k: None))
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    loaders = {n: DeviceCocoLoader(a.batch * n, a.batch, seed=1, device=dev) for n in (20, 60)}

    def phase(n, graph):
        eng.server_graph = bool(graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train(loaders[n])
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    phase(20, 0)                                   # libraries, allocator
    out = {'batch': a.batch, 'dim': a.dim, 'tag': a.tag}
    for rep in range(a.passes):
        e20, e60 = phase(20, 0), phase(60, 0)
        g20, g60 = phase(20, 1), phase(60, 1)
        out[f'pass{rep}'] = {'eager_ms_per_batch': round(e60 / 60 * 1e3, 2), 'eager_slope_ms': round((e60 - e20) / 40 * 1e3, 2),
                             'graph_ms_per_batch_60': round(g60 / 60 * 1e3, 2), 'graph_slope_ms': round((g60 - g20) / 40 * 1e3, 2),
                             'graph_fixed_cost_s': round(g20 - 20 * (g60 - g20) / 40, 3),
                             'graph_stats': dict(eng.graph_stats.get('train') or {})}
    out['mem_gb'] = {'allocated': round(torch.cuda.memory_allocated() / 2 ** 30, 2), 'reserved': round(torch.cuda.memory_reserved() / 2 ** 30, 2)}
    out['log'] = [m for m in msgs if 'GraphedStep' in m][:3]
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
