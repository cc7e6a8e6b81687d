"""Where the HOST time of a server step goes (the step is host-bound at the reference's public batch of 128): cProfile over N
steps with autograd's worker thread switched off, so that the backward's Python (custom Functions, ctypes calls) lands in the same
profile; prints the step time, the host issue time (no device wait inside) and the top functions by own time."""
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
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--top', type=int, default=45)
    a = ap.parse_args()
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(a.batch, dev, seed=1234, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.train_step(images, b[1], b[2], b[3])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3

    run(8)
    out = {'batch': a.batch}
    out['two_threads'] = dict(zip(('ms_per_step', 'host_issue_ms_per_step'), run(a.steps)))
    torch.autograd.set_multithreading_enabled(False)
    run(3)
    out['one_thread'] = dict(zip(('ms_per_step', 'host_issue_ms_per_step'), run(a.steps)))
    pr = cProfile.Profile()
    pr.enable()
    prof = run(a.steps)
    pr.disable()
    out['profiled'] = dict(zip(('ms_per_step', 'host_issue_ms_per_step'), prof))
    print(json.dumps(out))
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_stats(a.top)
    txt = s.getvalue()
    print('\n'.join(line[:170] for line in txt.splitlines()))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(a.top)
    print('\n'.join(line[:170] for line in s.getvalue().splitlines()))


if __name__ == '__main__':
    main()
