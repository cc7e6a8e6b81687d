"""Build-defined command-line flags, next to the reference's own (src/main.py:38-105, which stay untouched).

The reference driver builds its parser in `args()` (main.py:38-105) and hands the Namespace to `MMFL(args, wandb)`
(main.py:112-118).  Everything this package adds is OPTIONAL and read through `flags.get(args, name)`, so the unmodified
reference Namespace works; a driver that wants the extras on its command line adds one call:

    parser = argparse.ArgumentParser()          # main.py:39
    ...
    from creamfl_amd.flags import add_build_flags
    add_build_flags(parser)

Defaults keep the reference's semantics: `--server_dp 0` runs the server phases (global contrastive training, KD) on the
full batch on every rank -- bit-for-bit the single-process round, full-batch BatchNorm statistics -- and data-parallel
server phases (per-shard BatchNorm statistics, as in any data-parallel BatchNorm model) are an opt-in speed-up.
"""

# name -> (argparse keyword arguments, what it does)
BUILD_FLAGS = {
    'server_dp': (dict(type=int, default=0, choices=[0, 1]),
                  'multi-GPU: 1 = the server phases (global contrastive training, KD distillation) run data-parallel -- every '
                  'rank encodes 1/W of each public batch, features are all-gathered for the full-batch loss, gradients are '
                  'bucket-averaged; BatchNorm batch statistics are then per shard and the running statistics are averaged over '
                  'the ranks before evaluation / checkpointing.  0 (default) = replicated: the reference semantics'),
    'rep_wire': (dict(type=str, default='fp32', choices=['fp32', 'bf16']),
                 'multi-GPU: wire format of the public-set representations in the one all-gather per round'),
    'bucket_mb': (dict(type=int, default=32),
                  'multi-GPU: size of a gradient all-reduce bucket in MB (overlap granularity of the encoder-gradient all-reduce)'),
    'cnn_type': (dict(type=str, default=None), 'override the server image trunk (resnet18 / resnet50 / resnet101 / vit_b16)'),
    'bert_name': (dict(type=str, default=None), 'override the server text trunk (bert-mini / bert-base / bert-large)'),
    'image_size': (dict(type=int, default=224), 'side of the synthetic images when no loaders are passed'),
    'test_pairs': (dict(type=int, default=5000), 'captions of the synthetic test set when no loaders are passed'),
    'quiet': (dict(action='store_true'), 'no console logging'),
    'client_graph': (dict(type=int, default=1, choices=[0, 1]),
                     'capture the client contrast step (fixed B, M, D) in a HIP graph'),
    'mm_client_graph': (dict(type=int, default=1, choices=[0, 1]),
                        'the multi-modal client\'s contrast step (both towers, old model, A3 + A4, backward, clip + fused AdamP) in a HIP '
                        'graph too (needs --client_graph 1; the AdamP step count lives on the device inside the graph)'),
    'server_graph': (dict(type=int, default=0, choices=[0, 1]),
                     'single process only: the server\'s contrastive step and KD step (retrieval_trainer.py:192-214, MMFL.py:346-391) '
                     'replayed from HIP graphs, captured once per phase and round (the step is host-bound at the public batch 128)'),
    'client_channels_last': (dict(type=int, default=1, choices=[0, 1]),
                             'image encoders of the clients (ResNet client net, the multi-modal client\'s image tower) in channels_last '
                             'memory format: the reference\'s fp32 arithmetic on the library\'s NHWC convolutions and the fused fp32 BatchNorm '
                             'kernels (image client 28.6 -> 21.1 ms, multi-modal 37.6 -> 30.4 ms per contrast step on an MI355X); '
                             '0 = NCHW as the reference lays them out'),
    'client_bf16': (dict(type=int, default=0, choices=[0, 1]),
                    '1 = bf16 autocast for the clients\' image encoders (3.3 x faster contrast steps); BELOW the reference\'s client '
                    'precision (fp32, src/algorithms/ClientTrainer.py has no mixed precision), hence opt-in'),
    'client_conv_x3': (dict(type=int, default=1, choices=[0, 1]),
                       '1 (default) = the 3 x 3 / stride-1 convolutions of the clients\' fp32 channels_last image encoders (16 of '
                       'ResNet-18\'s 20) on csrc/conv3x3_x3.hip: forward and data gradient as 3 x bf16-split products on the bf16 matrix '
                       'pipe -- 16 mantissa bits per operand, outputs within 5e-6 of scale of fp64 (the library\'s fp32 kernels: 5e-7; '
                       'TF32, cuDNN\'s default for fp32 convolutions on A100-class GPUs: 10 bits); the client trains to the same '
                       'weights within the bounds the layout change is held to (tests/test_gpu_framework.py); weight gradient on the '
                       'library.  Image client 20.7 -> 15.8 ms per contrast step.  0 = every convolution on the library\'s fp32 kernels'),
    'miopen_immediate': (dict(type=int, default=0, choices=[0, 1]),
                         '1 = MIOpen immediate mode (no solver timing in the first step of every process: seconds instead of ~1 min); '
                         'only for the convolution shapes the shipped / recorded find-db holds -- other shapes silently get a '
                         'fallback kernel.  Same as CFL_MIOPEN_IMMEDIATE=1'),
}


def add_build_flags(parser):
    """Register every build-defined flag on an argparse parser (flags the parser already has are left alone)."""
    have = {a.dest for a in parser._actions}
    for name, (kw, doc) in BUILD_FLAGS.items():
        if name not in have:
            parser.add_argument('--' + name, help=doc, **kw)
    return parser


def get(args, name):
    """The value of a build-defined flag: the Namespace's, else the flag's default (the reference's own Namespace has none)."""
    kw = BUILD_FLAGS[name][0]
    default = kw.get('default', False if kw.get('action') == 'store_true' else None)
    return getattr(args, name, default)
