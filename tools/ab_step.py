"""Same-process A/B of the bench step: one engine, one batch, the two settings of a run-time knob alternating in rounds of timed
steps (A B A B ...), so that box, clocks, MIOpen choices and allocator state are shared -- separate `bench.py` processes on the
pool's boxes differ by several per cent from run to run, more than most of the effects worth measuring.

    python tools/ab_step.py --knob bres [--rounds 4] [--steps 10]

knobs:  bres   the gradient-join data-gradient GEMM on the B-resident streaming kernel (default) vs the tile kernel
        join   the gradient join of the residual blocks in the data-gradient GEMM's epilogue (default) vs in the BatchNorm backward
        wgfuse   conv3's weight gradient inside the BatchNorm backward-apply pass (default) vs the library's kernel on the side stream
        bnslice  the BatchNorm passes on the channel-sliced block map (no `final` launches, default) vs the whole-row map
        wgrad1   the 1 x 1 weight gradients on csrc/wgrad1x1.hip (default) vs the library's batched GEMM
        w1wgsN   that kernel aiming at N workgroups vs its default of 128
        w1hwN    that kernel for maps up to N x N only vs for every map (product default: 28)
        flushN   streams.FLUSH_POLICY = N vs 0 (when the queued weight gradients go to their stream)
        bertpack the BERT tower on the batch's real tokens only (default) vs on the reference's padded [B, L] frame
        wgrad3   the 3 x 3 weight gradients of layers 3 / 4 on csrc/wgrad3x3.hip (default) vs MIOpen's igemm_wrw
        w3splitN that kernel on N image ranges (N / 32 of the chip for the layer3 shape) vs its default of 256 workgroups
        wgrad    BOUND, not a product switch: every trunk weight gradient computed (default) vs replaced by a zero fill -- what the
                 side stream's 15 ms of library kernels cost the step (their own time is hidden; the contention is not)
        text     BOUND: the text tower run (default) vs its output replaced by a constant -- the image tower + head alone
        sides    BOUND: both of the above together -- the main stream with nothing beside it
"""
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
    ap.add_argument('--knob', default='bres')
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256)
    args = ap.parse_args()
    from creamfl_amd import _lib
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    cfg = default_config(embed_dim=512, cnn_type='resnet101', not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    eng.to_half()
    eng.model.train()
    b = coco_batch(args.batch, dev, seed=1234, bert=True)
    images = b[0].contiguous(memory_format=torch.channels_last)

    def set_knob(on):
        if args.knob == 'bres':
            lib.cfl_gemm_bf16_bres_min_m(32768 if on else (1 << 30))
        elif args.knob.startswith('dgradlib'):
            from creamfl_amd import ops
            ops.DGRAD_PLAIN_LIB[0] = int(args.knob[8:] or 64) if on else 0
        elif args.knob.startswith('flush'):
            from creamfl_amd import streams
            torch.cuda.synchronize()
            streams.FLUSH_POLICY[0] = int(args.knob[5:]) if on else 0
        elif args.knob == 'convstats':
            from creamfl_amd import ops
            ops.CONV_STATS[0] = bool(on)
        elif args.knob == 'wgfuse':
            from creamfl_amd import ops
            ops.WGRAD_FUSE[0] = bool(on)
        elif args.knob == 'bnslice':
            lib.cfl_bn_sliced(1 if on else 0)
        elif args.knob in ('wgrad', 'text', 'sides'):
            from creamfl_amd import ops
            if args.knob in ('wgrad', 'sides'):
                if not hasattr(ops, '_real_conv_wgrad'):
                    ops._real_conv_wgrad = ops._conv_wgrad
                ops._conv_wgrad = ops._real_conv_wgrad if on else (lambda a: torch.zeros_like(a[2]))
            if args.knob in ('text', 'sides'):
                m = eng.model
                if not hasattr(m, '_real_text_tower'):
                    m._real_text_tower = m._text_tower
                    with torch.no_grad(), torch.autocast('cuda', dtype=eng.autocast_dtype, enabled=eng.autocast_dtype is not None):
                        m._const_text = {k: (v.detach().clone() if torch.is_tensor(v) else v)
                                         for k, v in m._real_text_tower(b[1], b[2], b[3]).items()}
                m._text_tower = m._real_text_tower if on else (lambda *a, **k: dict(m._const_text))
        elif args.knob == 'wgrad1':
            from creamfl_amd import ops
            ops.WGRAD1[0] = bool(on)
        elif args.knob.startswith('w1hw'):       # 1 x 1 weight gradients on csrc/wgrad1x1.hip only up to this map height (off: every map)
            from creamfl_amd import ops
            ops.WGRAD1_MAX_HW[0] = int(args.knob[4:] or 14) if on else 0        # (off = every map; the product default is 28)
        elif args.knob.startswith('w1wgs'):
            lib.cfl_conv1x1_wgrad_workgroups(int(args.knob[5:] or 256) if on else 128)
        elif args.knob == 'wgrad3':
            from creamfl_amd import ops
            ops.WGRAD3[0] = bool(on)
        elif args.knob.startswith('w3split'):
            lib.cfl_conv3x3_wgrad_splits(int(args.knob[7:] or 16) if on else 0)
        elif args.knob == 'bertpack':     # the BERT tower on the batch's real tokens (default) vs on the padded [B, L] frame
            from creamfl_amd.networks.models import pcme as _pc
            _pc._NO_BERT_PACK[0] = not on
        elif args.knob == 'join':
            from creamfl_amd import ops
            ops._NO_JOIN_FUSE = not on
        else:
            raise SystemExit('unknown knob')

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.train_step(images, b[1], b[2], b[3])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    run(8)
    res = {'on': [], 'off': []}
    for r in range(args.rounds):
        for on in ((True, False) if r % 2 == 0 else (False, True)):
            set_knob(on)
            run(2)
            res['on' if on else 'off'].append(round(run(args.steps), 3))
    set_knob(True)
    mean = {k: round(sum(v) / len(v), 3) for k, v in res.items()}
    print(json.dumps({'knob': args.knob, 'ms_per_step': res, 'mean': mean, 'on_minus_off_ms': round(mean['on'] - mean['off'], 3)}))


if __name__ == '__main__':
    main()
