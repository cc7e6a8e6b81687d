#!/usr/bin/env python3
"""Headline benchmark: image-text pairs/sec of the server contrastive step.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N ...`: WORLD_SIZE / RANK / LOCAL_RANK come from the environment and WORLD_SIZE must
equal N), or -- plain `python bench.py --gpus N` -- this script re-executes itself under torch.distributed.run
(`launch_command`).  `--backend gloo` is the single-GPU smoke mode of the N > 1 path: the ranks share the visible GPU(s) and
collectives travel through host memory (tests/test_gpu_multirank.py uses the same transport); its line says so.

Workload at N = 1 = BASELINE.json configs[1]: "Server-only MSCOCO contrastive training, ResNet101 +
BERT-base, d=512, batch 256, 1 x MI355X" on MSCOCO-shaped synthetic batches resident in HBM.  One step =
PCME forward (ResNet-101 + PIE head | BERT-base + linear) -> MCSoftContrastiveLoss (HIP) -> backward ->
clip_grad_norm_(2) -> AdamP.step.  At N > 1 every rank runs the same per-GPU batch (weak scaling) as
large-batch global contrast: RCCL all-gather of the per-rank features, full-batch pair loss, DDP bucketed
all-reduce of the encoder gradients.

Prints ONE JSON line (rank 0) with the contract keys plus
  roofline      -- the dominant hand-written kernel of the step, timed with HIP events (start / stop events of the launch) inside the timed region
  cpu_baseline  -- the oracle port of the same step timed on the host cores (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Library set-up = the PRODUCT's (creamfl_amd/runtime.py: MIOpen find mode 2 + the find-db / kernel cache recorded on an
# MI355X, GPU_MAX_HW_QUEUES=8); importing the package applies the environment half before torch / HIP initialise, and
# TrainerEngine.create() switches cudnn.benchmark on.  Nothing here that a `src/main.py` user would not get.
from creamfl_amd import runtime as _runtime

_runtime.configure_env()

import torch

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (~6.3 TB/s achievable)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md


def algorithmic_cost(kernel, N, P, Cd, dh, D, n_f32=0, n_bf16=0):
    """(bound, algorithmic bytes or flops per launch) for each hand-written kernel at this workload
    (SURVEY section 8d)."""
    f = 4
    table = {
        # n_f32 fp32 parameters (fp32 grads) + n_bf16 bf16 trunk weights (bf16 grads, fp32 master, bf16 shadow)
        'cfl_adamp_pass1_kernel': ('hbm', 24 * n_f32 + 22 * n_bf16),    # read p,g,m,v; write m,v
        'cfl_adamp_pass3_kernel': ('hbm', 16 * n_f32 + 18 * n_bf16),    # read p,m,v; write p (+ shadow)
        'cfl_gradnorm_kernel': ('hbm', (4 * n_f32 + 2 * n_bf16) // 2),  # two launches (partial + final) share the id
        'cfl_pie_scores_kernel': ('hbm', N * P * dh * f + N * P * f),                       # read H, write scores
        'cfl_pie_pool_kernel': ('hbm', N * P * Cd * f + 2 * N * Cd * f + N * P * f),        # read X, write pooled+mean
        'cfl_pie_bwd_ds_kernel': ('hbm', N * P * Cd * f + N * Cd * f + N * P * f),          # read X, d_pooled
        'cfl_pie_bwd_dx_kernel': ('hbm', N * P * Cd * f + 2 * N * Cd * f),                  # write dX
        'cfl_pie_bwd_dh_kernel': ('hbm', 2 * N * P * dh * f),                               # read H, write dH
        'cfl_pie_bwd_dw2_kernel': ('hbm', (N * P // 64 + 1) * dh * f),
        'cfl_pie_epi_fwd_kernel': ('hbm', 5 * N * D * f),
        'cfl_pie_epi_bwd_kernel': ('hbm', 6 * N * D * f),
        'cfl_pie_epi_bwd_ln_kernel': ('hbm', 3 * N * D * f),
        'cfl_l2norm_fwd_kernel': ('hbm', 2 * N * D * f),
        'cfl_l2norm_bwd_kernel': ('hbm', 3 * N * D * f),
        'cfl_pair_prep_kernel': ('hbm', 2 * N * D * f),
        'cfl_pair_fwd_kernel': ('mfma', 2 * N * N * D),
        'cfl_pair_final_kernel': ('hbm', 4 * N * f),
        'cfl_pair_bwd_kernel': ('mfma', 4 * N * N * D),
    }
    return table.get(kernel)


def forward_flops(eng, images, captions, words, lens):
    """Model FLOPs of ONE forward of the server model on this batch (2 FLOP per MAC), by tower: every nn.Conv2d module by a
    forward hook, every linear layer -- modules AND the functional calls on a module's weight (the BERT tower's fused path runs
    `F.linear(x, weight)` on the concatenated Q/K/V, output and intermediate weights; the PIE w_1 projection) -- by wrapping
    torch.nn.functional.linear for the duration of the pass, plus the two matmuls that are neither (BERT's QK^T and PV).
    Returns {'image': ..., 'text': ..., 'total': ...}.  Used for the model-FLOPs utilisation of the step:
    step = 3 x forward.  (Round 4 hooked nn.Linear modules only: the fused BERT path never calls them, and the line reported
    12.2 instead of ~15 TFLOP per step.)"""
    import torch.nn as nn
    import torch.nn.functional as F
    tower = ['image']
    total = {'image': 0, 'text': 0}

    def conv_hook(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        total[tower[0]] += 2 * o.numel() * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]

    def enter(name):
        def pre(mod, inp):
            tower[0] = name
        return pre
    hs = [m.register_forward_hook(conv_hook) for m in eng.model.modules() if isinstance(m, nn.Conv2d)]
    hs.append(eng.model.img_enc.register_forward_pre_hook(enter('image')))
    hs.append(eng.model.txt_enc.register_forward_pre_hook(enter('text')))
    if getattr(eng.model, 'linear', None) is not None:
        hs.append(eng.model.linear.register_forward_pre_hook(enter('text')))        # the 768 -> D projection of the [CLS] state
    real_linear = F.linear

    def counting_linear(x, weight, bias=None):
        out = real_linear(x, weight, bias)
        total[tower[0]] += 2 * out.numel() * weight.shape[1]
        return out
    F.linear = counting_linear
    try:
        with torch.no_grad(), torch.autocast('cuda', dtype=eng.autocast_dtype, enabled=eng.autocast_dtype is not None):
            eng.model(images, captions, words, lens)
    finally:
        F.linear = real_linear
        for h in hs:
            h.remove()
    N = images.shape[0]
    bc = getattr(eng.model.txt_enc, 'config', None)
    if bc is not None:                                                               # BERT attention: QK^T and PV
        L = int(captions.shape[1])
        host = getattr(lens, '_cfl_host_lens', None)
        from creamfl_amd.networks.models import pcme as _pcme
        if host and not _pcme._NO_BERT_PACK[0] and max(host) <= 32 and images.is_cuda:
            # packed tower: a sequence attends over its own tokens only (the last, [CLS]-only layer over the padded frame: L keys)
            pairs = (bc.num_hidden_layers - 1) * sum(n * n for n in host) + N * L
            total['text'] += 4 * pairs * bc.hidden_size
        else:
            total['text'] += bc.num_hidden_layers * 4 * N * L * L * bc.hidden_size
    torch.cuda.synchronize()
    total['total'] = total['image'] + total['text']
    return total


def coco_1k_recall(dim, dev, seed=4321, noise=6.0):
    """COCO-1K retrieval protocol (eval_coco.py:336-390: 5 folds of 1000 images x 5000 captions, R@K averaged over the
    folds) on SYNTHETIC l2-normalised features -- image i, captions normalise(image_i + noise * unit) -- through the
    product evaluator (fp64 rank-count kernel, csrc/rank.hip).  Fold 0 is re-ranked by the CPU oracle: ranks must be equal."""
    import numpy as np
    import oracle
    from creamfl_amd.algorithms.eval_coco import COCOEvaluator
    ev = COCOEvaluator(eval_method='matmul', verbose=False, eval_device=str(dev), extract_device=str(dev), n_crossfolds=5)
    g = torch.Generator().manual_seed(seed)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g), dim=-1)
    r1 = {'i2t': [], 't2i': []}
    exact = True
    for fold in range(5):
        img = unit(1000, dim)
        cap = torch.nn.functional.normalize(img.repeat_interleave(5, 0) + noise * unit(5000, dim), dim=-1)
        icls, ccls = np.arange(1000), np.arange(5000) // 5
        r1['i2t'].append(ev.evaluate_recall(img, cap, icls, ccls)['recall_1'])
        r1['t2i'].append(ev.evaluate_recall(cap, img, ccls, icls)['recall_1'])
        if fold == 0:
            from creamfl_amd import ops
            got = ops.rank_count(img.to(dev), cap.to(dev), torch.as_tensor(icls), torch.as_tensor(ccls)).cpu().numpy()
            want = oracle.recall_ranks_count(img.numpy(), cap.numpy(), icls, ccls)
            exact = bool(np.array_equal(got.astype(np.float64), want))
    return {'i2t': round(float(np.mean(r1['i2t'])), 3), 't2i': round(float(np.mean(r1['t2i'])), 3),
            'protocol': 'COCO-1K: mean over 5 folds of 1000 images x 5000 captions', 'features': 'synthetic, d=%d, caption = '
            'normalize(image + %.1f * unit noise)' % (dim, noise), 'ranks_equal_cpu_oracle_fold0': exact}


# profiler id (runtime.hip) -> kernel names the offline PMC summary aggregates under (tools/pmc_summary.py)
PMC_ALIASES = {'cfl_gemm_bf16_kernel': ('cfl_gemm_bf16_nt_kernel', 'cfl_gemm_bf16_nt_bres_kernel')}


def offline_traffic(kernel, per_step, profiles_dir=None):
    """(HBM bytes per launch, provenance) of `kernel` from the offline PMC profile of this bench step under profiles/ -- quoted
    ONLY if that profile saw the same number of launches of the kernel per step as this run (a file taken before a fusion changed
    the launch count describes another kernel mix: round 2 paired 104-launch traffic with 103-launch algorithmic bytes); else
    (None, reason).  One profiler id can cover several kernel templates (PMC_ALIASES): their aggregates are combined,
    launch-weighted."""
    profiles_dir = profiles_dir or os.path.join(ROOT, 'profiles')
    source = 'none: no offline PMC profile matches this run (traffic = null)'
    for fn in ('r6_pmc_bench_traffic.json', 'r5_pmc_bench_traffic.json', 'r4_pmc_bench_traffic.json', 'r3_pmc_bench_traffic.json', 'r2_pmc_bench_traffic.json'):
        try:
            pmc = json.load(open(os.path.join(profiles_dir, fn)))
        except (OSError, ValueError):
            continue
        ent = pmc.get(kernel)
        if not ent:
            parts = [pmc[k] for k in PMC_ALIASES.get(kernel, ()) if k in pmc and pmc[k].get('launches_per_step')]
            if parts:
                n_all = sum(p['launches_per_step'] for p in parts)
                ent = {'launches_per_step': n_all,
                       'traffic_bytes': int(sum(p['traffic_bytes'] * p['launches_per_step'] for p in parts) / n_all)}
        if not ent:
            continue
        lps = ent.get('launches_per_step')
        if lps is None or abs(lps - per_step) >= 0.5:            # (the profiler may miss one launch of a run)
            source = ('none: profiles/%s was taken at %s launches of this kernel per step, this run has %g (traffic = null)'
                      % (fn, lps, per_step))
            continue
        return ent['traffic_bytes'], ('OFFLINE: profiles/%s (separate rocprofv3 --pmc passes over this bench step, %g launches per '
                                      'step as here; not measured in this run)' % (fn, per_step))
    return None, source


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_command(n_ranks, argv, port=None, python=None):
    """The command that starts `n_ranks` ranks of this script on ONE node (one process per GPU), argv = the script's own
    arguments, passed through unchanged.  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(n_ranks)),
            '--master-addr', '127.0.0.1', '--master-port', str(port or free_port()),
            os.path.abspath(__file__)] + list(argv)


def resolve_world(args, environ):
    """(world, rank, local_rank, must_spawn) from --gpus and the launcher's environment.  --gpus N > 1 without a launcher
    environment means: start the ranks ourselves; a launcher environment that disagrees with --gpus is an error (the line's
    n_gpus must be what the caller asked for)."""
    if 'WORLD_SIZE' not in environ:
        return (args.gpus, 0, 0, args.gpus > 1)
    world = int(environ['WORLD_SIZE'])
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    return (world, int(environ.get('RANK', '0')), int(environ.get('LOCAL_RANK', '0')), False)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL over xGMI, one GPU per rank (the measurement); gloo = smoke mode of the multi-rank path '
                         'on however many GPUs are visible (ranks share them, collectives through host memory)')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--no-alone', action='store_true', help='skip the 4 extra steps that measure the dominant kernel with the auxiliary streams off')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (pairs)')
    ap.add_argument('--dim', type=int, default=512)
    ap.add_argument('--cnn', default='resnet101')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'],
                    help='encoder trunk precision (reference: apex O2 fp16); head + loss are always fp32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dp', action='store_true', help='exercise the multi-GPU code path even with one rank')
    ap.add_argument('--bucket-mb', type=int, default=32,
                    help='gradient all-reduce bucket size (MB): ~10 buckets for 310 MB of gradients, all but the last overlap backward')
    ap.add_argument('--cpu-batch', type=int, default=0, help='batch of the CPU baseline (0 = the config\'s batch; falls back to 64 when '
                                                            'a step takes longer than --cpu-step-limit or memory is short)')
    ap.add_argument('--cpu-steps', type=int, default=10, help='timed CPU steps (median reported; BASELINE.md: >= 10)')
    ap.add_argument('--cpu-warmup', type=int, default=3)
    ap.add_argument('--cpu-step-limit', type=float, default=25.0)
    ap.add_argument('--cpu-timeout', type=int, default=540)
    ap.add_argument('--cpu-baseline-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--config', type=int, default=1, choices=[1, 2, 3],
                    help='BASELINE.json config: 1 = server step at 256 pairs per GPU (default, the metric\'s config); 2 = the client '
                         'side: contrast step of an image / text / multi-modal client and one full round, one client per GPU '
                         '(bench_clients.py); 3 = large-batch global contrast, 512 pairs per GPU (N = 4096 over 8 GPUs)')
    import bench_clients
    bench_clients.add_arguments(ap)
    ap.add_argument('--no-recall', action='store_true')
    ap.add_argument('--no-hot-kernels', action='store_true',
                    help='config 1, one rank: skip the `extra.hot_kernels` record (con_w at M = 50 000 and the pair loss at N = 4096, '
                         'timed after the step\'s timed region)')
    ap.add_argument('--no-client-steps', action='store_true',
                    help='config 1, one rank: skip the `extra.client_steps` record (the clients\' contrast steps of configs[2], measured by a '
                         'child process AFTER the timed region)')
    ap.add_argument('--client-steps-timeout', type=int, default=600)
    ap.add_argument('--no-mfu', action='store_true', help='skip the FLOP-counting forward pass (profiling runs: every launch then belongs to a step)')
    ap.add_argument('--prewarm', action='store_true',
                    help='let a CHILD process run 3 untimed steps first (MIOpen compiles / selects its kernels there).  Off by '
                         'default since round 4: the product does no such thing, and with the recorded find-db + kernel cache of '
                         'creamfl_amd/runtime.py the first process on a fresh box is within 0.5 %% of the second '
                         '(profiles/r4_first_process.txt)')
    ap.add_argument('--no-prewarm', action='store_true', help='(accepted for older command lines; the pre-warm is opt-in now)')
    ap.add_argument('--prewarm-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--watchdog', type=int, default=-1,
                    help='seconds after which a run that has not printed its line dumps every thread\'s stack and exits (a hung '
                         'collective must not eat the node: default 900 with more than one rank, off with one; 0 = off)')
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.config == 3:
        args.batch = 512
    if args.cpu_batch <= 0:
        args.cpu_batch = min(args.batch, 256)
    return args


def prewarm(args, local_rank, wait_s=400.0):
    """OPT-IN (--prewarm).  Rounds 2-3 measured the FIRST process on a fresh box 3-5 % slower than every later one, whatever its
    warm-up count (48.9 ms per step vs 46.5 / 46.3): MIOpen compiled kernels and fixed solver choices while that process ran.
    With the find-db of the final kernel mix recorded and the compiled-kernel cache shipped (creamfl_amd/runtime.py) the effect is
    gone -- fresh box, same lease: first 44.50 ms, second 44.28, caches wiped again 44.33 / 44.22 (profiles/r4_first_process.txt)
    -- so the default run is now exactly what a `src/main.py` user gets.  The child (3 untimed steps, no JSON, once per box and
    shape, a marker file makes later invocations skip it) stays available for A/B runs."""
    import subprocess
    import tempfile
    if not args.prewarm or args.no_prewarm or args.prewarm_child or not torch.cuda.is_available():
        return
    # ONE marker per node, whatever launched the ranks (the find-db directory is per LOCAL_RANK when a launcher set that
    # variable before this script was imported: a marker in there would be invisible to the other ranks)
    mark_dir = os.path.join(tempfile.gettempdir(), 'creamfl_prewarm_%d' % os.getuid())
    os.makedirs(mark_dir, exist_ok=True)
    mark = os.path.join(mark_dir, 'prewarmed_%s_%d_%d_%s' % (args.cnn, args.batch, args.dim, args.dtype))
    if os.path.exists(mark):
        return
    if local_rank != 0:
        # one child per node is enough (the compiled kernels and the find-db are shared through MIOPEN_USER_DB_PATH): the other
        # ranks wait for rank 0's marker, bounded -- a failed child must not hold the job
        t0 = time.perf_counter()
        while not os.path.exists(mark) and not os.path.exists(mark + '.failed') and time.perf_counter() - t0 < wait_s:
            time.sleep(min(1.0, wait_s / 4))
        return
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'GROUP_RANK', 'ROLE_RANK',
              'LOCAL_WORLD_SIZE', 'ROLE_WORLD_SIZE'):
        env.pop(k, None)
    if os.path.exists(mark + '.failed'):
        os.remove(mark + '.failed')                    # a new attempt: the other ranks wait for its outcome
    vis = [v for v in env.get('HIP_VISIBLE_DEVICES', '').split(',') if v != '']
    env['HIP_VISIBLE_DEVICES'] = vis[0] if vis else '0'
    cmd = [sys.executable, os.path.abspath(__file__), '--prewarm-child', '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch',
           str(args.batch), '--dim', str(args.dim), '--cnn', args.cnn, '--dtype', args.dtype, '--no-cpu-baseline', '--no-recall',
           '--no-alone']
    t0 = time.perf_counter()
    rc = subprocess.call(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.stderr.write('bench.py: library pre-warm child rc=%d, %.1f s\n' % (rc, time.perf_counter() - t0))
    try:
        open(mark if rc == 0 else mark + '.failed', 'w').write('ok\n' if rc == 0 else 'rc=%d\n' % rc)
    except OSError:
        pass


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.cpu_client_child:
        import bench_clients
        return bench_clients.cpu_client_child(args)
    world, rank, local_rank, must_spawn = resolve_world(args, os.environ)
    if must_spawn:
        import subprocess
        env = _runtime.child_env()                             # one find-db directory per LOCAL_RANK (no two processes on one text db)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes needs it on this driver)
        env.setdefault('OMP_NUM_THREADS', '4')
        raise SystemExit(subprocess.call(launch_command(args.gpus, sys.argv[1:]), env=env))
    wd = args.watchdog if args.watchdog >= 0 else (900 if world > 1 else 0)
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    prewarm(args, local_rank)
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL prints its version block at
    # communicator creation) are sent to stderr; the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in the product path)')
    n_visible = torch.cuda.device_count()
    if args.backend == 'nccl' and world > n_visible:
        raise SystemExit('bench.py: --gpus %d over RCCL needs %d GPUs, %d visible (use --backend gloo to smoke-run the '
                         'multi-rank path on fewer GPUs)' % (world, world, n_visible))
    dev_index = local_rank % n_visible
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        import datetime
        limit = datetime.timedelta(seconds=300)              # a collective that waits longer than this is a bug, not a straggler
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=limit)

    from creamfl_amd import _lib
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    _lib.load()
    if args.config == 2:
        # the client side of a round: contrast step per client kind + one MMFL round, one client per rank (bench_clients.py)
        import bench_clients
        bench_clients.run(args, world, rank, dev, use_dp, json_out)
        if use_dp:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if wd > 0:
            faulthandler.cancel_dump_traceback_later()
        return

    torch.manual_seed(1234)                       # identical initial weights on every rank
    cfg = default_config(embed_dim=args.dim, cnn_type=args.cnn, not_bert=False)
    eng = TrainerEngine(device=dev)
    eng.create(cfg, {'<pad>': 0}, None, False)
    eng.model_to_device()
    if args.dtype == 'bf16':
        eng.to_half()
    if use_dp:
        eng.enable_data_parallel(bucket_cap_mb=args.bucket_mb)
    eng.model.train()

    batch = coco_batch(args.batch, dev, seed=1234 + rank, bert=True)
    images, captions, words, lens = batch[0], batch[1], batch[2], batch[3]
    if args.dtype == 'bf16':
        images = images.contiguous(memory_format=torch.channels_last)

    def step():
        return eng.train_step(images, captions, words, lens)

    # what the text tower runs on: the batch's real tokens (packed, when the lengths are known on the host) or the padded frame
    from creamfl_amd.networks.models import pcme as _pcme
    _host_lens = getattr(lens, '_cfl_host_lens', None)
    text_tokens = {'padded_frame': int(captions.shape[0] * captions.shape[1]),
                   'real': int(sum(_host_lens)) if _host_lens else None,
                   'tower_runs_on': 'real tokens (packed, BertModel.pack_plan)' if (_host_lens and not _pcme._NO_BERT_PACK[0]
                                                                                   and max(_host_lens) <= 32) else 'padded frame'}

    def fence():
        if use_dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    from creamfl_amd import ops as _ops
    fwd_flops = None if args.no_mfu else forward_flops(eng, images, captions, words, lens)
    # Warm-up.  Its last step is event-timed for EVERY hand-written kernel: that gives the per-kernel table and
    # tells which kernel dominates.  In the timed region only that one kernel is timed (its launches go through
    # hipExtLaunchKernelGGL with a start / stop event: the dispatch's own timestamps, no marker packets in the stream).
    for i in range(args.warmup):
        last = (i == args.warmup - 1)
        if last:
            torch.cuda.synchronize()
            _lib.prof_select(None)
            _lib.prof_reset()
            _lib.prof_enable(True)
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    warm_prof = _lib.prof_query() if args.warmup > 0 else {}
    # the dominant hand-written kernel = largest share of the step among the kernels with an algorithmic cost model
    modelled = [k for k in warm_prof if k.startswith('cfl_bn_') or k == 'cfl_gemm_bf16_kernel'
                or algorithmic_cost(k, 1, 1, 8, 4, 8, 1, 1)]
    dominant = max(modelled, key=lambda k: warm_prof[k][1]) if modelled else 'cfl_bn_bwd_apply_kernel'
    fence()
    for k_ in _ops.BN_COUNTERS:
        _ops.BN_COUNTERS[k_] = 0
    _ops.GEMM_COUNTERS['flops'] = _ops.GEMM_COUNTERS['bytes'] = 0
    _lib.prof_reset()
    _lib.prof_select(dominant)
    if not os.environ.get('BENCH_NO_PROF'):      # (diagnosis only: the timed region without the dominant kernel's event timing -> no roofline)
        _lib.prof_enable(True)
    t0 = time.perf_counter()
    trace = []
    for _ in range(args.steps):
        loss, _ = step()
        trace.append(loss.detach())              # device scalars; read after the timed region
    fence()
    dt = time.perf_counter() - t0
    if args.prewarm_child:
        return
    _lib.prof_enable(False)
    _lib.prof_select(None)
    prof = _lib.prof_query()
    loss_val = float(loss.detach())
    if not all(bool(torch.isfinite(t)) for t in trace):
        raise SystemExit('bench.py: non-finite loss in the timed region -- the measurement is invalid')
    if os.environ.get('BENCH_TRACE'):
        sys.stderr.write('loss per timed step: %s\n' % ' '.join('%.4f' % float(t) for t in trace))

    if use_dp:
        t = torch.tensor([dt], device=dev if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    if rank == 0:
        # roofline of the dominant hand-written kernel (by total time in the timed region)
        Cd = eng.model.img_enc.cnn_dim
        roof = None
        cand = []
        Nloss = args.batch * world
        n_bf16 = sum(p.numel() for p in eng.model.parameters() if p.dtype == torch.bfloat16)
        n_f32 = sum(p.numel() for p in eng.model.parameters() if p.dtype == torch.float32) + 2
        # fused BN kernels run once per BatchNorm layer with a different shape each: their algorithmic bytes are
        # accumulated over the launches of the timed region (bf16 = 2 B / element) and divided per launch
        def kernel_cost(base, n):
            """(bound, algorithmic bytes or FLOP per launch) of a hand-written kernel over the launches just profiled"""
            bc = _ops.BN_COUNTERS
            bn_bytes = {
                'cfl_bn_stats_kernel': 2 * (bc['fwd'] - bc['fwd_pre']),      # (statistics the producing GEMM's epilogue took over)
                'cfl_bn_apply_kernel': 4 * bc['fwd'] + 2 * bc['fwd_res'] + bc['fwd_mask'] // 8,   # + the 1-bit ReLU mask
                # bwd reduce: read dy (+ second upstream gradient), x and the ReLU mask bits (with a residual; otherwise
                # the mask comes from x); bwd apply: the same reads, write dx (+ the residual gradient)
                'cfl_bn_bwd_reduce_kernel': 4 * (bc['bwd'] + bc['bwd_wg']) + bc['bwd_relu'] // 8 + 2 * bc['bwd_two'],   # (bwd_wg: layers whose apply pass is the fused weight-gradient kernel)
                'cfl_bn_bwd_apply_kernel': 6 * bc['bwd'] + bc['bwd_relu'] // 8 + 2 * bc['bwd_res'] + 2 * bc['bwd_two'],
            }
            cost = algorithmic_cost(base, args.batch, 49, Cd, Cd // 2, args.dim, n_f32, n_bf16)
            if base.startswith('cfl_pair_'):
                cost = algorithmic_cost(base, Nloss, 49, Cd, Cd // 2, args.dim)
            if base in bn_bytes and bn_bytes[base] > 0:
                cost = ('hbm', bn_bytes[base] / n)
            if base == 'cfl_gemm_bf16_kernel' and _ops.GEMM_COUNTERS['flops'] > 0:
                # 1x1-convolution data gradients: K, N <= 2048 puts them below the ridge point (2500 TFLOP/s / 8 TB/s =
                # 312 FLOP/B), i.e. the binding roof is HBM unless the FLOP time is the larger one
                gf, gb = _ops.GEMM_COUNTERS['flops'] / n, _ops.GEMM_COUNTERS['bytes'] / n
                cost = ('mfma_bf16', gf) if gf / (BF16_MFMA_PEAK_TFLOPS * 1e12) > gb / (HBM_PEAK_GBPS * 1e9) else ('hbm', gb)
            return cost

        for name, (n, ms) in prof.items():
            cost = kernel_cost(name, n)
            if cost:
                cand.append((ms, name, n, cost))
        if cand:
            ms, name, n, (bound, work) = max(cand)
            us = ms / n * 1e3
            if bound == 'hbm':
                ach = work / (us * 1e-6) / 1e9
                roof = {'kernel': name, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBPS,
                        'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBPS, 4), 'traffic': None,
                        'avg_launch_us': round(us, 2), 'launches': n, 'algorithmic_bytes': int(work),
                        'share_of_step_ms': round(ms / args.steps, 3)}
            else:
                ach = work / (us * 1e-6) / 1e12
                peak = BF16_MFMA_PEAK_TFLOPS if bound == 'mfma_bf16' else F32_MFMA_PEAK_TFLOPS
                roof = {'kernel': name, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak,
                        'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': None,
                        'avg_launch_us': round(us, 2), 'launches': n, 'algorithmic_flops': work,
                        'share_of_step_ms': round(ms / args.steps, 3)}
        if roof is not None and args.batch == 256 and args.cnn == "resnet101" and args.dtype == "bf16":
            # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this timed run, so the value
            # measured offline at this exact shape (separate rocprofv3 --pmc passes, see the file) is reported -- but only if
            # that profile saw the SAME number of launches per step as this run (a file taken before a fusion changed the
            # launch count describes another kernel mix: round 2 paired 104-launch traffic with 103-launch algorithmic bytes).
            roof['traffic'], roof['traffic_source'] = offline_traffic(roof['kernel'], roof['launches'] / float(args.steps))
        if roof is not None and not (os.environ.get('CFL_NO_TWO_STREAM') and os.environ.get('CFL_NO_SIDE_WGRAD')):
            # the text tower and the convolution weight gradients run on auxiliary HIP streams: part of these launches
            # share HBM with their kernels, so the per-launch rate is a lower bound of what the kernel reaches alone
            # (CFL_NO_TWO_STREAM=1 CFL_NO_SIDE_WGRAD=1 measures that)
            roof['concurrent_stream'] = True
            if world == 1 and not args.no_alone:
                # the same kernel with the auxiliary streams switched off (3 extra steps, outside the timed region)
                from creamfl_amd.networks import backbones as _bb
                from creamfl_amd.networks.models import pcme as _pc
                saved = (_pc._NO_TWO_STREAM, _bb._NO_SIDE_WGRAD)
                _pc._NO_TWO_STREAM, _bb._NO_SIDE_WGRAD = True, True
                try:
                    step()
                    fence()
                    for k_ in _ops.BN_COUNTERS:
                        _ops.BN_COUNTERS[k_] = 0
                    _ops.GEMM_COUNTERS['flops'] = _ops.GEMM_COUNTERS['bytes'] = 0
                    _lib.prof_reset()
                    _lib.prof_select(roof['kernel'])
                    _lib.prof_enable(True)
                    for _ in range(3):
                        step()
                    fence()
                    _lib.prof_enable(False)
                    _lib.prof_select(None)
                    n1, ms1 = _lib.prof_query()[roof['kernel']]
                    bound1, work1 = kernel_cost(roof['kernel'], n1)
                    us1 = ms1 / n1 * 1e3
                    ach1 = work1 / (us1 * 1e-6) / (1e9 if bound1 == 'hbm' else 1e12)
                    roof['alone'] = {'achieved': round(ach1, 1), 'frac': round(ach1 / roof['peak'], 4),
                                     'avg_launch_us': round(us1, 2), 'launches': n1,
                                     'how': 'same step with the text-tower and weight-gradient streams off'}
                finally:
                    _pc._NO_TWO_STREAM, _bb._NO_SIDE_WGRAD = saved
        hip_us = {k: round(ms / n * 1e3, 2) for k, (n, ms) in sorted((warm_prof or prof).items())}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, args)
        recall = None
        if not args.no_recall:
            recall = coco_1k_recall(args.dim, dev)
        extra = None
        if world == 1 and args.config == 1 and not args.no_client_steps:
            extra = {'client_steps': client_steps(args)}
        if world == 1 and args.config == 1 and not args.no_hot_kernels:
            extra = dict(extra or {})
            extra['hot_kernels'] = hot_kernels(dev)
        # model-FLOPs utilisation of the step: forward + backward = 3 x forward model FLOPs per step and GPU
        mfma_peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == 'bf16' else F32_MFMA_PEAK_TFLOPS
        mfu = None
        if fwd_flops is not None:
            step_tflop = 3.0 * fwd_flops['total'] / 1e12
            per = lambda f: round(3.0 * f / 1e12 / (ms_per_step * 1e-3) / mfma_peak, 4)
            mfu = {'model_tflop_per_step_per_gpu': round(step_tflop, 3), 'gflop_per_pair': round(3.0 * fwd_flops['total'] / args.batch / 1e9, 2),
                   'achieved_tflops_per_gpu': round(step_tflop / (ms_per_step * 1e-3), 1), 'peak_tflops': mfma_peak,
                   'mfu': per(fwd_flops['total']),
                   # the towers run CONCURRENTLY on two streams: each share is that tower's FLOPs over the WHOLE step time
                   'by_tower': {'image': {'tflop_per_step': round(3.0 * fwd_flops['image'] / 1e12, 3), 'mfu': per(fwd_flops['image'])},
                                'text': {'tflop_per_step': round(3.0 * fwd_flops['text'] / 1e12, 3), 'mfu': per(fwd_flops['text'])}},
                   'how': '3 x forward FLOPs (nn.Conv2d by hooks; every linear layer incl. the functional calls of the fused BERT path '
                          'and PIE w_1 by wrapping F.linear; + BERT QK^T, PV) / step time / dense MFMA peak'}

        comm = None
        if use_dp:
            # what one step puts on the wire per rank: the bucketed encoder-gradient all-reduce and the two feature all-gathers
            comm = dict(eng.dp.reducer.comm_stats())
            comm['bucket_mb'] = args.bucket_mb
            comm['gather_bytes_per_step'] = 2 * args.batch * world * args.dim * 4       # image + caption features, fp32, all ranks' rows
            comm['bucket_bytes'] = comm['bucket_bytes'][:16]
        out = {
            'metric': 'image-text pairs/sec (contrastive step)', 'value': round(value, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': 'server contrastive step: ResNet101+BERT-base PCME, d=%d, per-GPU batch %d, '
                                   'MCSoftContrastiveLoss + clip + AdamP (BASELINE.json configs[%d])' % (args.dim, args.batch, args.config),
                       'global_batch': args.batch * world, 'cnn': args.cnn, 'text': 'bert-base',
                       'encoder_precision': args.dtype, 'head_loss_precision': 'f32', 'text_tokens': text_tokens,
                       'parallelism': 'dp%d' % world, 'loss': round(loss_val, 4)},
            'ranks': {'world_size': torch.distributed.get_world_size() if use_dp else 1,
                      'backend': ('rccl' if args.backend == 'nccl' else 'gloo (SMOKE MODE: not a scaling measurement)') if use_dp
                      else None, 'rccl_ranks': torch.distributed.get_world_size() if (use_dp and args.backend == 'nccl') else 0,
                      'gpus_visible': n_visible},
            'comm': comm, 'roofline': roof, 'cpu_baseline': cpu, 'mfu': mfu, 'recall_1': recall, 'extra': extra,
            'parity_unpinned': ['AdamP (adamp==0.3.0 is not vendored: checked against the paper restatement oracle/adamp.py)'],
            'hip_kernels_us_warmup_step': hip_us,
        }
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    if use_dp:
        torch.distributed.barrier()           # rank 0 is still writing its line (recall evaluation): tear down together
        torch.distributed.destroy_process_group()
    if wd > 0:
        faulthandler.cancel_dump_traceback_later()


def usable_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    reports 256 logical CPUs but grants a 16-CPU quota; oversubscribing it stalls the baseline)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _host_memory_limit_gb():
    """What this process may allocate: the cgroup limit if there is one, else MemAvailable."""
    try:
        v = open('/sys/fs/cgroup/memory.max').read().strip()
        if v != 'max':
            return int(v) / 2 ** 30
    except (OSError, ValueError):
        pass
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                return int(line.split()[1]) / 2 ** 20
    except (OSError, ValueError):
        pass
    return None


def hot_kernels(dev):
    """`extra.hot_kernels` of the default line: the two hot-path kernels of BASELINE.json's north_star that the configs[1] step does not
    run at their full size -- con_w (row A5: log-probabilities of ONE client's [M, D] representations against the global ones, M = 50 000,
    the public set of configs[2]) at D = 256 / 512 and the pair loss (row A1) forward + backward at the global batch of configs[3],
    N = 4096, d = 512 -- timed with events on the current stream AFTER the step's timed region (nothing of it can touch `value`).
    FLOPs are the algorithmic 2 M^2 D (con_w) and 3 x 2 N^2 D (pair loss); the roof is the 3 x bf16-split rate of the dense bf16 MFMA
    peak (every fp32 product = three bf16 MFMAs)."""
    import torch.nn.functional as F
    try:
        from creamfl_amd import ops
        roof = BF16_MFMA_PEAK_TFLOPS / 3.0
        g = torch.Generator(device=dev).manual_seed(7)
        unit = lambda n, d: F.normalize(torch.randn(n, d, generator=g, device=dev), dim=-1)

        def timed(fn, iters, warm=2):
            for _ in range(warm):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        out = {'how': 'con_w: torch events around 4 calls after 2 warm-ups on the current stream; pair loss: the sum of its kernels\' times (HIP events per launch, 10 calls) beside the wall time of the call; after the timed region'}
        M = 50000
        for D in (256, 512):
            G = unit(M, D)
            V = F.normalize(G + 0.5 * unit(M, D), dim=-1)
            ms = timed(lambda: ops.conw_logprob(V, G), 4)
            tf = 2.0 * M * M * D / (ms * 1e-3) / 1e12
            out['conw_logprob_M50000_D%d' % D] = {'ms_per_client': round(ms, 3), 'algorithmic_tflops': round(tf, 1),
                                                  'of_3xbf16_roof': round(tf / roof, 3)}
            del G, V
        N, D = 4096, 512
        I = unit(N, D).requires_grad_(True)
        T = F.normalize(I.detach() + 0.5 * unit(N, D), dim=-1).requires_grad_(True)
        a = torch.tensor([15.0], device=dev, requires_grad=True)
        b = torch.tensor([15.0], device=dev, requires_grad=True)

        def step():
            loss, _ = ops.pair_loss(I, T, a, b)
            loss.backward()
        ms = timed(step, 10)
        # the call is 6 launches under autograd: in this process (hundreds of modules, the backward on the calling thread) its WALL time is
        # host time; the kernels' own time comes from the library's per-kernel profiler (two HIP events per launch) in a second pass
        from creamfl_amd import _lib
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        prof = {k: (n, t) for k, (n, t) in _lib.prof_query().items() if k.startswith('cfl_pair_')}
        kus = sum(t for _, t in prof.values()) / 10 * 1e3
        tf = 3 * 2.0 * N * N * D / (kus * 1e-6) / 1e12
        out['pair_loss_fwd_bwd_N4096_D512'] = {'kernels_us_per_call': round(kus, 1), 'wall_us_per_call': round(ms * 1e3, 1),
                                               'kernels': {k: round(t / n * 1e3, 2) for k, (n, t) in sorted(prof.items())},
                                               'algorithmic_tflops': round(tf, 1), 'of_3xbf16_roof': round(tf / roof, 3),
                                               'pairs_per_s_of_the_kernels': round(N / (kus * 1e-6))}
        return out
    except Exception as e:                                   # never let an extra record take the line down
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}


def client_steps(args):
    """`extra.client_steps` of the default line: the CLIENT side of configs[2] where the driver can see
    it -- the contrast step of one image / text / multi-modal client at the config-2 defaults (public batch 128 of 224 x 224 images /
    COCO-shaped captions, banks M = 50 000, D = 256, inter + intra; the product path = replayed from a HIP graph, eager beside it), and
    the bank pass (rows A3 + A4, the kernel north_star names) HIP-event timed inside the eager region with its fraction of HBM.
    Measured by a CHILD process (`bench.py --config 2 --round none`) after this process's timed region and after its GPU memory was
    released: nothing of it can touch `value`.  The configs[1] fields of the line are untouched."""
    import subprocess
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.abspath(__file__), '--config', '2', '--round', 'none', '--steps', '30', '--warmup', '5',
           '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.client_steps_timeout)
    except subprocess.TimeoutExpired:
        return {'error': 'client-steps child exceeded %d s' % args.client_steps_timeout}
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    if res.returncode != 0 or not lines:
        return {'error': 'client-steps child rc=%d: %s' % (res.returncode, res.stderr.decode()[-300:])}
    line = json.loads(lines[-1])
    out = {'how': 'child process `bench.py --config 2 --round none --steps 30 --warmup 5` after the timed region; fp32 clients (the '
                  'reference\'s client precision), product path = HIP-graph replay',
           'workload': line['config']['workload'], 'child_seconds': round(time.perf_counter() - t0, 1)}
    for kind, name in (('img', 'image'), ('txt', 'text'), ('mm', 'multi_modal')):
        c = (line.get('clients') or {}).get(kind)
        if not c:
            continue
        best = c.get('graph') or c['eager']
        out[name] = {'ms_per_step': best['ms_per_step'], 'pairs_per_s': best['pairs_per_s'], 'path': 'hip graph' if c.get('graph') else 'eager',
                     'eager_ms_per_step': c['eager']['ms_per_step'], 'batch': c['batch'],
                     'capture_failed': (c.get('graph') or {}).get('capture_failed'),
                     'a3a4_kernels_us_per_step': c.get('a3a4_kernels_us_per_step'),
                     'hand_written_kernels_us_per_step': c.get('hand_written_kernels_us_per_step'),
                     'bank_pass_avg_launch_us': (c.get('bank_pass') or {}).get('avg_launch_us')}
    roof = line.get('roofline')
    if roof:
        out['bank_pass_roofline'] = {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us',
                                                              'launches', 'algorithmic_bytes', 'how')}
    # the same steps with every convolution on the library's fp32 kernels (`--client_conv_x3 0`): the image encoders' 3 x 3 convolutions
    # run by default as 3 x bf16-split products (csrc/conv3x3_x3.hip: 16 mantissa bits per operand) -- both forms belong in the line
    out['convolutions'] = '3 x 3 / stride 1: csrc/conv3x3_x3.hip (3 x bf16 split, fp32-class); the rest: library fp32'
    try:
        res = subprocess.run(cmd + ['--client-conv-x3', '0', '--only-kinds', 'img,mm'], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=args.client_steps_timeout)
        lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
        if res.returncode == 0 and lines:
            lib = json.loads(lines[-1])
            out['with_library_fp32_convolutions'] = {
                name: ((lib['clients'][kind].get('graph') or lib['clients'][kind]['eager'])['ms_per_step'])
                for kind, name in (('img', 'image'), ('mm', 'multi_modal')) if kind in (lib.get('clients') or {})}
    except subprocess.TimeoutExpired:
        out['with_library_fp32_convolutions'] = {'error': 'timeout'}
    out['child_seconds'] = round(time.perf_counter() - t0, 1)
    return out


def cpu_baseline(cfg, args):
    """the CPU restatement of the SAME step on the GPU box's host cores -- identical seeded inputs (the
    config's batch, 256), threads = all usable cores, >= 3 warm-ups, median of >= 10 timed steps; core count, CPU model string and
    thread count in the line.  Runs in a CHILD process after the GPU timing (it cannot perturb the timed region, and its memory
    is gone before the recall evaluation).  If one step at the config's batch takes longer than ~25 s, or the host cannot hold its
    ~60 GB of fp32 activations, the child falls back to batch 64 and says so."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', '--batch', str(args.batch), '--dim', str(args.dim),
           '--cnn', args.cnn, '--cpu-batch', str(args.cpu_batch), '--cpu-steps', str(args.cpu_steps), '--cpu-warmup',
           str(args.cpu_warmup)]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    note = ''
    for attempt in range(2):
        try:
            res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.cpu_timeout)
        except subprocess.TimeoutExpired:
            return {'value': None, 'error': 'cpu baseline child exceeded %d s' % args.cpu_timeout}
        lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
        if res.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out['sample'] += note
            return out
        if res.returncode == CPU_CHILD_RSS_EXIT and attempt == 0 and args.cpu_batch > 64:
            # the child's memory watchdog stopped it before the host ran short: once more at batch 64
            note = '; batch %d stopped by the memory watchdog -> batch 64' % args.cpu_batch
            cmd[cmd.index('--cpu-batch') + 1] = '64'
            continue
        return {'value': None, 'error': 'cpu baseline child rc=%d: %s' % (res.returncode, res.stderr.decode()[-300:])}


CPU_CHILD_RSS_EXIT = 43


def _rss_watchdog(limit_gb):
    """The CPU port at batch 256 holds ~60 GB of fp32 activations.  Never let it push the host into its memory limit (an OOM
    kill takes the box with it): a thread polls this process's resident set and exits the process at 60 % of what it may
    use; the parent then repeats the baseline at batch 64."""
    import threading
    page = os.sysconf('SC_PAGE_SIZE')
    budget = 0.6 * limit_gb * 2 ** 30

    def poll():
        while True:
            try:
                rss = int(open('/proc/self/statm').read().split()[1]) * page
            except (OSError, ValueError):
                return
            if rss > budget:
                os._exit(CPU_CHILD_RSS_EXIT)
            time.sleep(0.25)
    threading.Thread(target=poll, daemon=True).start()


def cpu_baseline_child(args):
    """The oracle port of the step on the host cores (the child process of cpu_baseline)."""
    import oracle.step as ostep
    from oracle.adamp import AdamP as OracleAdamP
    from creamfl_amd.networks.models import get_model
    from creamfl_amd.utils.config import default_config
    from creamfl_amd.utils.synthetic import coco_batch
    from types import SimpleNamespace
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = default_config(embed_dim=args.dim, cnn_type=args.cnn, not_bert=False)
    mem = _host_memory_limit_gb()
    if mem is not None:
        _rss_watchdog(mem)
    note = ''
    same_batch = None
    for batch in ([args.cpu_batch] if args.cpu_batch <= 64 else [args.cpu_batch, 64]):
        if batch > 64 and mem is not None and mem < 0.35 * batch:        # ~0.25 GB of saved fp32 activations per ResNet-101 sample
            note = '; batch %d needs ~%d GB of host memory, %.0f GB available -> batch 64' % (batch, int(0.25 * batch), mem)
            continue
        torch.manual_seed(1234)
        model = get_model({'<pad>': 0}, cfg.model, False).train()
        crit = SimpleNamespace(negative_scale=torch.nn.Parameter(torch.tensor([15.0])),
                               shift=torch.nn.Parameter(torch.tensor([15.0])))
        params = [p for p in model.parameters() if p.requires_grad] + [crit.negative_scale, crit.shift]
        opt = OracleAdamP(params, lr=cfg.optimizer.learning_rate, weight_decay=cfg.optimizer.weight_decay)
        b = coco_batch(batch, 'cpu', seed=1234, bert=True)        # the GPU run's rank-0 batch (same seed, same generator)

        def one():
            t0 = time.perf_counter()
            ostep.contrastive_step_cpu(model, crit, opt, b, cfg.train.grad_clip)
            return time.perf_counter() - t0
        warm = [one(), one()]
        if batch > 64 and warm[1] > args.cpu_step_limit:
            note = '; one step at batch %d took %.1f s (> %.0f s) -> batch 64' % (batch, warm[1], args.cpu_step_limit)
            # the GPU's own workload: the second step at the config's batch IS a measurement -- one warm-up, one
            # timed step -- and is reported beside the full protocol at batch 64 instead of being thrown away
            same_batch = {'batch': batch, 'value': round(batch / warm[1], 3), 'unit': 'pairs/s', 'step_s': round(warm[1], 3),
                          'first_step_s': round(warm[0], 3), 'protocol': '1 warm-up + 1 timed step (a step takes > %.0f s: the '
                          'median-of-%d protocol runs at batch 64)' % (args.cpu_step_limit, args.cpu_steps)}
            del model, opt, b
            continue
        warm += [one() for _ in range(max(0, args.cpu_warmup - 2))]
        times = sorted(one() for _ in range(max(1, args.cpu_steps)))
        med = times[len(times) // 2]
        print(json.dumps({
            'value': round(batch / med, 3), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port', 'threads': torch.get_num_threads(),
            'os_cpu_count': os.cpu_count(), 'cpu_model': _cpu_model_string(), 'batch': batch,
            'step_s_median': round(med, 3), 'step_s_min_max': [round(times[0], 3), round(times[-1], 3)],
            'protocol': 'BASELINE.md section 3: %d warm-ups + median of %d timed steps, threads = usable cores' % (len(warm), len(times)),
            'sample': 'same step (%s+BERT-base fp32, d=%d, oracle head+loss, clip, AdamP) on the same seeded batch, batch %d%s'
                      % (args.cnn, args.dim, batch, note),
            'same_batch_as_gpu': same_batch if same_batch is not None else ({'batch': batch, 'value': round(batch / med, 3),
                                                                             'unit': 'pairs/s', 'step_s': round(med, 3),
                                                                             'protocol': 'the full protocol above'}
                                                                            if batch == args.batch else None)}))
        return
    raise SystemExit('cpu baseline: no batch size fits')


if __name__ == '__main__':
    main()
