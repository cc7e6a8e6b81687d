#!/usr/bin/env python3
"""What clock does the chip hold under a dense-MFMA kernel?  Runs con_w log-probabilities (M = 50 000, D = 256: ~3.3 ms of
3 x bf16 MFMA work per call) back to back for ~2 s while a thread samples `rocm-smi --showclocks`, and the same for an HBM-streaming
kernel (BatchNorm forward).  The dense peaks of MI355X_MICROARCH.md are quoted at 2.4 GHz."""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import creamfl_amd  # noqa: E402,F401
import torch  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=5).stdout.decode()
            m = re.search(r'sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz', txt)
            p = re.search(r'Power \(W\):\s*([0-9.]+)', txt)
            out.append((int(m.group(1)) if m else None, float(p.group(1)) if p else None))
        except Exception:
            pass
        time.sleep(0.05)


def run(label, fn, seconds=2.0):
    fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out), daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    clk = [c for c, _ in out if c]
    pw = [p for _, p in out if p]
    return {'kernel': label, 'ms_per_call': round(dt / n * 1e3, 3), 'sclk_mhz_min_med_max': [min(clk), sorted(clk)[len(clk) // 2], max(clk)] if clk else None,
            'power_w_med': sorted(pw)[len(pw) // 2] if pw else None, 'samples': len(out)}


def main():
    from creamfl_amd import _lib, ops
    _lib.load()
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cuda').manual_seed(2)
    G = torch.nn.functional.normalize(torch.randn(50000, 256, generator=g, device=dev), dim=-1)
    V = torch.nn.functional.normalize(G + 0.5 * torch.randn(50000, 256, generator=g, device=dev), dim=-1)
    x = torch.randn(256, 256, 56, 56, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = [run('idle', lambda: None, 0.5),
           run('con_w log-probabilities (dense 3 x bf16 MFMA)', lambda: ops.conw_logprob(V, G)),
           run('bf16 elementwise stream (HBM-bound)', lambda: x.mul_(1.0))]
    if '--step' in sys.argv:                                       # the bench step (BASELINE configs[1]) under the same sampler
        from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
        from creamfl_amd.utils.config import default_config
        from creamfl_amd.utils.synthetic import coco_batch
        del x
        torch.manual_seed(1234)
        cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
        eng = TrainerEngine(device=dev)
        eng.create(cfg, {'<pad>': 0}, None, False)
        eng.model_to_device()
        eng.to_half()
        eng.model.train()
        b = coco_batch(256, dev, seed=1234, bert=True)
        images = b[0].contiguous(memory_format=torch.channels_last)
        for _ in range(4):
            eng.train_step(images, b[1], b[2], b[3])
        res.append(run('server contrastive step (R101 + BERT-base, batch 256)', lambda: eng.train_step(images, b[1], b[2], b[3]), 4.0))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
