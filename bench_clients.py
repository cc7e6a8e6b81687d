"""bench.py --config 2: the CLIENT side of a CreamFL round (BASELINE.json configs[2]: "Full CreamFL: 10 img + 10 txt + 5 mm
clients, con_w agg, 8 clients/round on 8 x MI355X"), the path `north_star` leads with for N GPUs:

    rank r trains client r (reference loop: src/algorithms/MMFL.py:226-247)
      -> its contrast loop against the frozen global banks (src/algorithms/ClientTrainer.py:369-429,
         src/algorithms/MMClientTrainer.py:150-224: encoder forward, old-model forward, A3 + A4, backward, SGD / AdamP)
      -> generate_logits into this rank's slice of the [W, K, M, D] gather buffer
      -> ONE all_gather_into_tensor (RCCL over xGMI)
      -> row-sharded con_w (MMFL.py:298-335) + all-gather of the aggregate rows
      -> KD distillation on the server model (MMFL.py:343-391)

Two measurements, one JSON line (rank 0):

  1. the CONTRAST STEP (the metric: image-text pairs/sec of the contrastive step).  Timed region = `--steps` calls of the
     trainer's own `contrast_step_fn` -- the function `tra()` / `train_epoch()` iterate -- on a device-resident public batch
     (B = 128, 224 x 224 images, COCO-shaped captions) against M = 50 000 random unit banks, D = 256, inter + intra,
     interintra_weight 0.5.  N = 1: one image client (ResNet-18 client net; replayed from a HIP graph, the product default, and
     eager), one text client (bi-GRU + PIE; graph and eager), one multi-modal client (PCME small: ResNet-18 + GRU,
     AdamP; graph and eager); `value` = the image client's graphed step.  N > 1: rank r times the step of the client KIND it would own
     (KINDS8), `value` = pairs of all ranks / slowest rank's time.
  2. ONE ROUND (`MMFL.train(0)`, product code, untouched: phases are timed by wrapping its methods from outside): global
     contrastive training, global representations, the sampled clients (local training + contrast loop + representations), the
     representation all-gather, con_w, KD, evaluation.  All loaders are synthetic and born in HBM.

`roofline` = the bank pass (`cfl_bank_stream_kernel`, rows A3 + A4: the kernel `north_star` names), HIP-event timed inside the
eager timed region; `cpu_baseline` = the oracle's port of the image client's contrast step on the host cores (child process).
"""
import copy
import json
import os
import random
import subprocess
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
KINDS8 = ('img', 'txt', 'mm', 'img', 'txt', 'img', 'txt', 'mm')          # client kind of rank r (10 : 10 : 5 ~ 3 : 3 : 2)
A34_KERNELS = ('cfl_bank_stream_kernel', 'cfl_lse_final_kernel', 'cfl_bank_bwd_kernel')
HBM_PEAK_GBPS = 8000.0


def add_arguments(ap):
    ap.add_argument('--only-kinds', default='img,txt,mm', help='single process: which clients\' steps to measure')
    ap.add_argument('--client-conv-x3', type=int, default=1, help='the clients\' fp32 3x3 convolutions on csrc/conv3x3_x3.hip (flags.client_conv_x3)')
    ap.add_argument('--round-first', type=int, default=1, help='the federation round before the clients\' micro-benchmarks (a real run\'s order)')
    ap.add_argument('--pub', type=int, default=50000, help='config 2: public-set size M (banks, representations, con_w)')
    ap.add_argument('--client-batch', type=int, default=128, help='config 2: public-loader batch B of the contrast loops')
    ap.add_argument('--client-dim', type=int, default=256, help='config 2: feature_dim D (src/main.py:103 default)')
    ap.add_argument('--image-size', type=int, default=224)
    ap.add_argument('--round', default='full', choices=['full', 'none'], help='config 2: also run and time one MMFL round')
    ap.add_argument('--round-pub', type=int, default=6400,
                    help='config 2: public-set size of the timed ROUND (0 = --pub).  The default is 50 public batches instead of 391: every '
                         'loop of a round is linear in the number of public batches (reported per batch too); what is NOT linear -- con_w '
                         'and the representation all-gather -- is timed separately at the full --pub (`full_M` in the line)')
    ap.add_argument('--round-warm', type=int, default=1, choices=[0, 1],
                    help='config 2: run a miniature round (2 public batches, 1 private batch per client) first: every convolution '
                         'problem of a round is then known to the libraries (PyTorch asks MIOpen for a timed search per new shape: '
                         '~1 min per network and batch size) and the timed round measures the round, not the searches')
    ap.add_argument('--server-cnn', default='resnet101')
    ap.add_argument('--server-bert', default='bert-base-uncased')
    ap.add_argument('--clients', default='10,10,5', help='config 2: image, text, multi-modal clients in the federation')
    ap.add_argument('--clients-per-round', type=int, default=8)
    ap.add_argument('--client-train-n', default='5000,12000,5800',
                    help='config 2: private samples per image / text / multi-modal client (CIFAR-100 / 10, AG_NEWS / 10, Flickr30k / 5)')
    ap.add_argument('--rep-wire', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--client-layout', default='channels_last', choices=['channels_last', 'nchw'],
                    help='config 2: memory format of the clients\' image encoders (fp32 either way; nchw = as the reference lays them out)')
    ap.add_argument('--client-bf16', type=int, default=0, choices=[0, 1],
                    help='config 2: bf16 autocast for the clients\' image encoders -- BELOW the reference\'s fp32 client precision: the '
                         'line then says dtype bf16 and is a companion number, not the configs[2] measurement')
    ap.add_argument('--cpu-client-child', action='store_true', help='(internal)')
    ap.add_argument('--server-graph', type=int, default=0, choices=[0, 1],
                    help='config 2: --server_graph of the federation (the server\'s contrastive and KD steps replayed from HIP graphs)')
    ap.add_argument('--mm-client-graph', type=int, default=1, choices=[0, 1],
                    help='config 2: --mm_client_graph of the federation (the multi-modal client\'s contrast step from a HIP graph)')


def reference_namespace(a, dev_index, M):
    """The Namespace src/main.py:38-105 hands to MMFL, with the values of configs[2] (+ the build's optional flags)."""
    ni, nt, nm = (int(x) for x in a.clients.split(','))
    return SimpleNamespace(
        name='/tmp/creamfl_bench_c2', feature_dim=a.client_dim, pub_data_num=M, not_bert=False, mlp_local=False, server_lr=2e-4,
        local_epochs=1, comm_rounds=1, num_img_clients=ni, num_txt_clients=nt, num_mm_clients=nm,
        client_num_per_round=a.clients_per_round, agg_method='con_w', contrast_local_intra=True, contrast_local_inter=True,
        interintra_weight=0.5, loss_scale=False, kd_weight=0.3, disable_distill=False, save_client=False, device=dev_index,
        cnn_type=a.server_cnn, bert_name=a.server_bert, image_size=a.image_size, test_pairs=5000 if M >= 5000 else max(100, M // 2),
        quiet=True, save_checkpoints=False, server_dp=0, rep_wire=a.rep_wire, client_graph=1,
        client_channels_last=int(a.client_layout == 'channels_last'), client_bf16=int(a.client_bf16),
        server_graph=int(a.server_graph), mm_client_graph=int(a.mm_client_graph), client_conv_x3=int(a.client_conv_x3))


def build_federation(a, dev, M, mini=False):
    """MMFL + its trainers + device-born loaders (public set, test set, every client's private set).  Every loader yields FULL
    batches only (sizes rounded to multiples of the batch): a ragged last batch is another convolution problem, i.e. another timed
    search of every layer.  mini: one private batch per client, a 5-batch test set."""
    from creamfl_amd.algorithms.MMFL import MMFL
    from creamfl_amd import dist as cdist
    from creamfl_amd.utils.synthetic import DeviceClientLoader, DeviceCocoLoader
    ns = reference_namespace(a, dev.index or 0, M)
    B, S = a.client_batch, a.image_size
    ns.test_pairs = 5 * 2 * B if mini else (20 * 2 * B if M >= 5000 else 5 * 2 * B)
    algo = MMFL(ns, None)
    algo.device = dev
    algo.config.dataloader.batch_size = B
    bs_uni = min(512, 4 * B)
    scale = min(1.0, M / 50000.0)                      # a reduced public set shrinks the private sets with it
    want = [int(x) for x in a.client_train_n.split(',')]
    n_img, n_txt, n_mm = [(bs if mini else max(bs, int(round(n * scale / bs)) * bs)) for n, bs in zip(want, (bs_uni, bs_uni, B))]
    loaders = {
        'img': [DeviceClientLoader('img', n_img, bs_uni, 100, dev, seed=11 + i, img=S) for i in range(ns.num_img_clients)],
        'txt': [DeviceClientLoader('txt', n_txt, bs_uni, 4, dev, seed=31 + i) for i in range(ns.num_txt_clients)],
        'mm': [DeviceCocoLoader(n_mm, B, seed=51 + i, bert=False, device=dev, img=S) for i in range(ns.num_mm_clients)],
    }
    algo.create_model(ns, client_loaders=loaders)
    algo.load_dataset(ns, dataloaders={
        algo._pub_key(False): DeviceCocoLoader(M, B, seed=1, device=dev, img=S),
        algo._pub_key(True): DeviceCocoLoader(M, 2 * B, seed=1, device=dev, img=S),
        'test': DeviceCocoLoader(ns.test_pairs, 2 * B, seed=2, device=dev, img=S, captions_per_image=5)})
    algo.client_sampler = lambda trainers, k: cdist.balanced_sample(trainers, k)
    return algo, {'private_samples': {'img': n_img, 'txt': n_txt, 'mm': n_mm}, 'test_pairs': ns.test_pairs}


def first_of_kind(algo, kind, owner=None, world=1):
    pool = {'img': algo.img_local_trainers, 'txt': algo.txt_local_trainers, 'mm': algo.mm_local_trainers}[kind]
    for t in pool:
        if owner is None or t.client_idx % world == owner:
            return t
    return pool[0]


def _fence(use_dist):
    if use_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def _wall(fn, steps, use_dist=False):
    _fence(use_dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    _fence(use_dist)
    return time.perf_counter() - t0, out


class StepHarness:
    """One client's contrast step on a resident public batch: the trainer's own `contrast_step_fn`, set up as `run()` /
    `tra()` / `train_epoch()` set it up (device, deep-copied old model in eval mode, phase switches), torn down the same way."""

    def __init__(self, trainer, kind, banks, batch, dev):
        self.t, self.kind, self.dev = trainer, kind, dev
        g_img, g_txt = banks
        images, captions, _, lens = batch[0], batch[1], batch[2], batch[3]
        B = images.shape[0]
        gen = torch.Generator().manual_seed(99)
        idx = torch.randperm(g_img.shape[0], generator=gen)[:B]
        self.d_idx_host, self.d_idx_dev = tuple(idx.tolist()), idx.to(dev)
        t = trainer
        t._to_device()                                   # device + the layout / precision the build flags ask for, as run() does
        if kind == 'mm':
            t.model.train()
            t.old_model = copy.deepcopy(t.model).eval()
            step = t.contrast_step_fn(g_img, g_txt, True, True)
            self.eager = lambda: step(images, captions, None, lens, self.d_idx_host)
            self.graph_fn, self.graph_in, self.graph_opt = None, None, None
            from creamfl_amd import flags, ops
            from creamfl_amd.algorithms.ClientTrainer import caption_graph_width, pad_captions
            from creamfl_amd.algorithms.optimizers import AdamP
            if (isinstance(t.optimizer, AdamP) and int(flags.get(t.args, 'mm_client_graph')) and t.model.config.not_bert
                    and ops.gru_last_supported(getattr(t.model.txt_enc, 'rnn', None))):
                # the product path since the fused AdamP reads its step count on the device (MMClientTrainer._graphed_for_round)
                width = caption_graph_width(captions.shape[1])
                self.graph_fn = lambda im, cap, ln, di: step(im, cap, None, ln, di)
                self.graph_in = (images, pad_captions(captions, width), torch.as_tensor(lens, dtype=torch.int64), self.d_idx_dev)
                self.graph_opt = t.optimizer
        else:
            t.model.train()
            t.old_model = copy.deepcopy(t.model).eval()
            for m in (t.model, t.old_model):
                m.phase, m.is_train = 'extract_conv_feature', False
            g_same, g_other = (g_img, g_txt) if kind == 'img' else (g_txt, g_img)
            step = t.contrast_step_fn(g_same, g_other, True, True)
            if kind == 'img':
                self.eager = lambda: step(images, None, None, self.d_idx_host)
                self.graph_fn = lambda im, di: step(im, None, None, di)
                self.graph_in = (images, self.d_idx_dev)
            else:
                self.eager = lambda: step(None, captions, lens, self.d_idx_host)
                self.graph_fn, self.graph_in = None, None
                from creamfl_amd import ops
                from creamfl_amd.algorithms.ClientTrainer import caption_graph_width, pad_captions
                if ops.gru_last_supported(getattr(t.model, 'rnn', None)):
                    # the product path of a text client since gru.hip: lengths on the device, every batch padded to one width
                    width = caption_graph_width(captions.shape[1])
                    self.graph_fn = lambda cap, ln, di: step(None, cap, ln, di)
                    self.graph_in = (pad_captions(captions, width), torch.as_tensor(lens, dtype=torch.int64), self.d_idx_dev)
        self.B = B

    def close(self):
        t = self.t
        if self.kind != 'mm':
            t.model.phase, t.model.is_train = 'None', True
        t.old_model = None


def measure_client(trainer, kind, banks, batch, dev, steps, warmup, use_dist=False, profile_steps=5):
    """Eager (+ graphed: image and text clients) wall time of the contrast step, the hand-written kernels' table from a profiled eager
    pass, and the bank pass HIP-event timed INSIDE the eager timed region (prof_select: only that kernel carries events)."""
    from creamfl_amd import _lib
    from creamfl_amd.graphs import GraphedStep
    h = StepHarness(trainer, kind, banks, batch, dev)
    out = {'kind': kind, 'batch': h.B}
    for _ in range(max(1, warmup)):
        loss = h.eager()
    torch.cuda.synchronize()
    # all hand-written kernels, per launch and per step (separate pass: every launch carries two events)
    _lib.prof_select(None)
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(profile_steps):
        h.eager()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    table = _lib.prof_query()
    out['hip_kernels'] = {k: {'launches_per_step': round(n / profile_steps, 2), 'us_per_launch': round(ms / n * 1e3, 2),
                              'us_per_step': round(ms / profile_steps * 1e3, 1)} for k, (n, ms) in sorted(table.items())}
    a34 = sum(table[k][1] for k in A34_KERNELS if k in table) / profile_steps * 1e3
    hand = sum(ms for _, ms in table.values()) / profile_steps * 1e3
    # the eager timed region, the bank pass event-timed in it
    _lib.prof_reset()
    _lib.prof_select('cfl_bank_stream_kernel')
    _lib.prof_enable(True)
    dt, loss = _wall(h.eager, steps, use_dist)
    _lib.prof_enable(False)
    _lib.prof_select(None)
    bank = _lib.prof_query().get('cfl_bank_stream_kernel')
    out['eager'] = {'ms_per_step': round(dt / steps * 1e3, 3), 'pairs_per_s': round(h.B * steps / dt, 1), 'seconds': dt}
    out['bank_pass'] = None if not bank else {'launches': bank[0], 'avg_launch_us': round(bank[1] / bank[0] * 1e3, 2)}
    if h.graph_fn is not None:
        gs = GraphedStep(h.graph_fn, warmup=3, optimizer=getattr(h, 'graph_opt', None))
        for _ in range(4 + max(0, warmup)):                       # 3 eager warm-ups, the capture (+ first replay), replays
            gs(*h.graph_in, device=dev)
        dtg, loss = _wall(lambda: gs(*h.graph_in, device=dev), steps, False)     # (local fences: the multi-modal client has no such region)
        out['graph'] = {'ms_per_step': round(dtg / steps * 1e3, 3), 'pairs_per_s': round(h.B * steps / dtg, 1), 'seconds': dtg,
                        'replays': gs.replays, 'capture_failed': gs.failed}
    best = out.get('graph') or out['eager']
    out['a3a4_kernels_us_per_step'] = round(a34, 1)
    out['hand_written_kernels_us_per_step'] = round(hand, 1)
    out['a3a4_share_of_step'] = {'eager': round(a34 / (out['eager']['ms_per_step'] * 1e3), 4),
                                 'product_path': round(a34 / (best['ms_per_step'] * 1e3), 4)}
    out['loss'] = round(float(loss), 4)
    h.close()
    return out


class PhaseClock:
    """Times the phases of MMFL.train from OUTSIDE (the product code is not touched): bound methods are wrapped on the instances,
    every wrapper fences the device on both sides."""

    def __init__(self):
        self.t, self.n = {}, {}
        self._undo = []

    def wrap(self, obj, name, phase):
        fn = getattr(obj, name)

        def timed(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                torch.cuda.synchronize()
                self.t[phase] = self.t.get(phase, 0.0) + time.perf_counter() - t0
                self.n[phase] = self.n.get(phase, 0) + 1
        setattr(obj, name, timed)
        self._undo.append((obj, name, fn))

    def restore(self):
        for obj, name, fn in reversed(self._undo):
            try:
                delattr(obj, name)                     # instance attribute shadowing the class's method
            except AttributeError:
                setattr(obj, name, fn)
        self._undo = []


def timed_round(algo, use_dist, round_n=0):
    from creamfl_amd import dist as cdist
    clk = PhaseClock()
    clk.wrap(algo.engine, 'train', 'global_train')
    clk.wrap(algo, 'extract_global_features', 'global_reps')
    for t in algo.total_local_trainers:
        clk.wrap(t, 'run', 'clients_train')
        clk.wrap(t, 'generate_logits', 'clients_reps')
    clk.wrap(algo, 'aggregation', 'con_w')
    clk.wrap(algo, 'distill', 'distill_total')
    clk.wrap(algo.engine, 'evaluate', 'evaluate')
    # class-level: the gather buffer is created inside train()
    orig_gather, orig_agc = cdist.RepGatherBuffer.gather, cdist.all_gather_cat
    comm = {'rep_all_gather_s': 0.0, 'agg_all_gather_s': 0.0, 'gather_bytes': 0, 'agg_gather_bytes': 0, 'collectives': 0}

    def gather(self):
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()                # the collective's own time, not the wait for the slowest client
        t0 = time.perf_counter()
        orig_gather(self)
        torch.cuda.synchronize()
        comm['rep_all_gather_s'] += time.perf_counter() - t0
        comm['gather_bytes'] += self.buf.numel() * self.buf.element_size()
        comm['collectives'] += 1

    def agc(t, group=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig_agc(t, group)
        torch.cuda.synchronize()
        comm['agg_all_gather_s'] += time.perf_counter() - t0
        comm['agg_gather_bytes'] += out.numel() * out.element_size()
        return out
    cdist.RepGatherBuffer.gather = gather
    random.seed(1234)
    _fence(use_dist)
    t0 = time.perf_counter()
    try:
        orig_aggregation = algo.aggregation            # (the wrapped one)

        def aggregation(i_vec, t_vec):
            cdist.all_gather_cat = agc
            try:
                return orig_aggregation(i_vec, t_vec)
            finally:
                cdist.all_gather_cat = orig_agc
        algo.aggregation = aggregation
        algo.train(round_n)
    finally:
        cdist.RepGatherBuffer.gather = orig_gather
        cdist.all_gather_cat = orig_agc
    _fence(use_dist)
    total = time.perf_counter() - t0
    clk.restore()
    ph = dict(clk.t)
    ph['kd'] = ph.get('distill_total', 0.0) - ph.get('con_w', 0.0)
    ph.pop('distill_total', None)
    ph['round_total'] = total
    ph['other'] = total - sum(v for k, v in ph.items() if k not in ('round_total',))
    kinds = ['mm' if t.__class__.__name__ == 'MMClientTrainer' else ('img' if 'img' in t.modalities else 'txt')
             for t in algo.cur_trainers]
    return ph, clk.n, comm, kinds


def full_size_exchange(a, dev, banks, world, rank, use_dist):
    """What does NOT scale with the number of public batches, at the full public-set size: the round's ONE representation
    all-gather (every rank contributes the blocks of one multi-modal client = 2 x [M, D], the worst case of a slot) and con_w over
    8 client representations per modality (row-sharded across the ranks + the all-gather of the aggregate rows)."""
    from creamfl_amd import dist as cdist
    M, D = banks[0].shape
    g = torch.Generator(device=dev).manual_seed(777 + rank)
    out = {'M': M, 'D': D, 'clients_per_modality': 8}
    if use_dist:
        plan = [[(r, ('img', 'txt'))] for r in range(world)]
        wire = torch.bfloat16 if a.rep_wire == 'bf16' else torch.float32
        buf = cdist.RepGatherBuffer(plan, M, D, dev, wire)
        for v in buf.out_views(0).values():
            v.copy_(torch.nn.functional.normalize(torch.randn(M, D, generator=g, device=dev), dim=-1))
        buf.gather()                                     # warm
        _fence(True)
        t0 = time.perf_counter()
        buf.gather()
        _fence(True)
        out['rep_all_gather_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
        out['gather_bytes'] = buf.buf.numel() * buf.buf.element_size()
        del buf
    gs = torch.Generator(device=dev).manual_seed(888)         # the same representations on every rank, as after the gather
    vecs = [torch.nn.functional.normalize(banks[0] + 0.5 * torch.randn(M, D, generator=gs, device=dev), dim=-1) for _ in range(8)]
    cdist.conw_aggregate_sharded(vecs, banks[1])         # warm (bank image build, workspace)
    _fence(use_dist)
    t0 = time.perf_counter()
    agg = cdist.conw_aggregate_sharded(vecs, banks[1])
    _fence(use_dist)
    out['con_w_ms_per_modality'] = round((time.perf_counter() - t0) * 1e3, 3)
    out['con_w_ms_per_client'] = round(out['con_w_ms_per_modality'] / 8, 3)
    out['con_w_flop_per_client'] = 2.0 * M * M * D
    out['agg_finite'] = bool(torch.isfinite(agg).all())
    return out


def cpu_client_child(a):
    """The oracle's port of the image client's contrast step on the host cores (bounded sample)."""
    import bench
    import oracle
    import oracle.step as ostep
    from creamfl_amd.networks.resnet_client import resnet18_client
    cores = bench.usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    B, D, M, S = a.client_batch, a.client_dim, a.pub, a.image_size
    model = resnet18_client(pretrained=False, num_class=100, is_train=True, scale=128, mlp_local=False, embed_dim=D)
    g = torch.Generator().manual_seed(5)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g), dim=-1)
    g_img, g_txt = unit(M, D), unit(M, D)
    images = torch.randn(B, 3, S, S, generator=g)
    idx = torch.randperm(M, generator=g)[:B].tolist()
    state = ostep.ClientStepState(model.state_dict(), lr=1e-4)
    one = lambda: ostep.client_contrast_step_cpu(state, images, g_img, g_txt, idx, interintra_weight=0.5)
    t0 = time.perf_counter()
    one()
    first = time.perf_counter() - t0
    n = 5 if first < 4.0 else 3
    times = []
    for _ in range(n):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    print(json.dumps({'value': round(B / med, 3), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port', 'threads': torch.get_num_threads(),
                      'cpu_model': bench._cpu_model_string(), 'step_s_median': round(med, 3),
                      'sample': 'image client contrast step (ResNet-18 client net fwd + bwd, old-model fwd, inter + intra against '
                                'M = %d banks, SGD), B = %d, D = %d, %d x %d images: 1 warm-up + median of %d steps'
                                % (M, B, D, S, S, n)}))


def cpu_baseline(a):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '2', '--cpu-client-child', '--pub', str(a.pub), '--client-batch',
           str(a.client_batch), '--client-dim', str(a.client_dim), '--image-size', str(a.image_size)]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    try:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=a.cpu_timeout)
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': 'cpu baseline child exceeded %d s' % a.cpu_timeout}
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    if res.returncode == 0 and lines:
        return json.loads(lines[-1])
    return {'value': None, 'error': 'cpu baseline child rc=%d: %s' % (res.returncode, res.stderr.decode()[-300:])}


def run(a, world, rank, dev, use_dist, json_out):
    """Called by bench.main() for --config 2 after the process group (if any) exists."""
    from creamfl_amd import _lib
    from creamfl_amd.utils.synthetic import coco_batch_on_device
    _lib.load()
    torch.manual_seed(1234)
    random.seed(1234)
    M, B, D, S = a.pub, a.client_batch, a.client_dim, a.image_size
    Mr = min(a.round_pub or M, M)
    Mr = max(2 * B, (Mr // (2 * B)) * (2 * B))            # whole batches of both public loaders (B and 2 B)
    algo, fed = build_federation(a, dev, Mr)
    g = torch.Generator(device=dev).manual_seed(4321)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g, device=dev), dim=-1)
    banks = (unit(M, D), unit(M, D))
    batch = coco_batch_on_device(B, dev, seed=1234 + rank, img=S)
    # ORDER (round 6): the federation round runs FIRST, as a real run has it -- a process that ran the three clients' micro-benchmarks
    # before the round showed its server phases ~10 ms per public batch slower with a long tail (35 GB reserved instead of 19:
    # tools/federation_step_trace.py, profiles/r5_federation_step_trace.jsonl); `--round-first 0` restores round 5's order.
    def round_part():
        rnd = None
        full = None
        if a.round == 'full':
            warm_s = None
            if a.round_warm:
                t0 = time.perf_counter()
                mini, _ = build_federation(a, dev, 4 * B, mini=True)
                random.seed(4321)
                mini.train(0)
                _fence(use_dist)
                warm_s = round(time.perf_counter() - t0, 1)
                del mini
                torch.cuda.empty_cache()
            # the federation's FIRST round starts on an empty allocator and a fresh server engine (its first steps allocate their
            # gigabytes: 44 ms per public batch where the same loop runs 28-35 ms warm, docs/history/tools/server_phase_probe.py); every later round
            # of a real run is the steady state, so the SECOND round is the one reported phase by phase, the first one beside it
            ph0, _, _, _ = timed_round(algo, use_dist, 0)
            ph, counts, comm, sampled = timed_round(algo, use_dist, 1)
            if use_dist:
                keys = sorted(ph)
                t = torch.tensor([ph[k] for k in keys], device=dev if a.backend == 'nccl' else 'cpu', dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                ph_max = {k: float(v) for k, v in zip(keys, t)}
            else:
                ph_max = ph
            n_pub_batches = -(-Mr // B)
            full = full_size_exchange(a, dev, banks, world, rank, use_dist)
            per_batch = {k: round(ph_max[k] / n_pub_batches * 1e3, 2) for k in ('global_train', 'global_reps', 'kd') if k in ph_max}
            rnd = {'pub_data_num': Mr, 'public_batches': n_pub_batches, 'clients_sampled': sampled,
                   'miniature_warm_up_round_s': warm_s, 'ms_per_public_batch': per_batch,
                   'ran_before_the_client_micro_benchmarks': bool(a.round_first),
                   'timed_round': 'the second round of the federation (steady state: warm allocator, warm server engine); the first '
                                  'round is in first_round_phases_s_rank0',
                   'first_round_phases_s_rank0': {k: round(v, 3) for k, v in sorted(ph0.items())},
                   'scaling_note': 'every loop of the round is linear in the public batches (391 at the full M = 50 000); con_w '
                                   '(quadratic in M) and the representation all-gather are in `full_M` at the full size',
                   'clients_trained_by_this_rank': counts.get('clients_train', 0),
                   'phases_s_rank0': {k: round(v, 3) for k, v in sorted(ph.items())},
                   'phases_s_max_over_ranks': {k: round(v, 3) for k, v in sorted(ph_max.items())},
                   'comm': {'gather_bytes': comm['gather_bytes'], 'rep_all_gather_ms': round(comm['rep_all_gather_s'] * 1e3, 3),
                            'rep_collectives': comm['collectives'], 'agg_gather_bytes': comm['agg_gather_bytes'],
                            'agg_all_gather_ms': round(comm['agg_all_gather_s'] * 1e3, 3),
                            'con_w_ms': round(ph.get('con_w', 0.0) * 1e3, 3), 'rep_wire': a.rep_wire},
                   'private_samples_per_client': fed['private_samples'],
                   'graphs': {'server_graph': int(a.server_graph), 'mm_client_graph': int(a.mm_client_graph),
                              'server': dict(getattr(algo.engine, 'graph_stats', {}) or {}),
                              'mm_clients': [g for g in (getattr(t, 'graph_stats', None) for t in algo.mm_local_trainers) if g][:2]},
                   'recall_1_after_round': None}
            try:
                sc = algo.best_scores['test']
                rnd['recall_1_after_round'] = {'i2t': sc['i2t']['recall_1'], 't2i': sc['t2i']['recall_1']}
            except (TypeError, KeyError):
                pass
        return rnd, full

    def client_part():
        kinds = tuple(k for k in ('img', 'txt', 'mm') if k in a.only_kinds.split(',')) if world == 1 else (KINDS8[rank % 8],)
        clients = {}
        for kind in kinds:
            tr = first_of_kind(algo, kind, rank if world > 1 else None, world)
            clients[kind] = measure_client(tr, kind, banks, batch, dev, a.steps, a.warmup, use_dist)
        mine = clients[kinds[0]]
        best = mine.get('graph') or mine['eager']
        dt, pairs = best['seconds'], float(B * a.steps)
        if use_dist:
            t = torch.tensor([dt, pairs], device=dev if a.backend == 'nccl' else 'cpu', dtype=torch.float64)
            tmax = t.clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
            dt, pairs = float(tmax[0]), float(t[1])
        value = pairs / dt
        # roofline of the bank pass, from this rank's eager timed region
        roof = None
        if mine.get('bank_pass'):
            us = mine['bank_pass']['avg_launch_us']
            work = M * D * 4 + 3 * B * D * 4                  # the bank (4 B per element: its bf16 hi / lo image) once + F, F_old, dF
            if kinds[0] == 'mm':
                work = M * D * 4 + 3 * B * D * 4              # (per launch: the multi-modal client launches it once per modality)
            ach = work / (us * 1e-6) / 1e9
            roof = {'kernel': 'cfl_bank_stream_kernel', 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': round(ach / HBM_PEAK_GBPS, 4), 'traffic': None, 'avg_launch_us': us, 'launches': mine['bank_pass']['launches'],
                    'algorithmic_bytes': int(work),
                    'how': 'HIP start / stop events of the launch inside the eager timed region of the %s client (the graph replays the '
                           'same kernel; events cannot ride in a captured graph)' % kinds[0]}
        return kinds, clients, dt, value, roof

    if a.round_first:
        rnd, full = round_part()
        kinds, clients, dt, value, roof = client_part()
    else:
        kinds, clients, dt, value, roof = client_part()
        rnd, full = round_part()
    if rank == 0:
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a)
        out = {
            'metric': 'image-text pairs/sec (contrastive step)', 'value': round(value, 2), 'unit': 'pairs/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16 (opt-in, below the reference\'s fp32 clients)' if a.client_bf16 else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'client contrast step (BASELINE.json configs[2]): %s client(s), public batch %d of %dx%d images / '
                                   'COCO-shaped captions resident in HBM, banks M = %d, D = %d, inter + intra (weight 0.5), encoder '
                                   'forward + old-model forward + A3/A4 + backward + optimizer; value = %s'
                                   % ('one image / text / multi-modal' if world == 1 else 'one per rank, kinds ' + ','.join(KINDS8[r % 8] for r in range(world)),
                                      B, S, S, M, D, 'the image client replayed from its HIP graph (product default)' if world == 1
                                      else 'pairs of all ranks / slowest rank'),
                       'global_batch': B * world, 'parallelism': 'clients%d' % world, 'client_image_layout': a.client_layout,
                       'client_nets': {'img': 'resnet18_client', 'txt': 'bi-GRU + PIE (language_model.EncoderText)',
                                       'mm': 'PCME small: ResNet-18 + GRU, AdamP'}},
            'ranks': {'world_size': world, 'backend': None if not use_dist else ('rccl' if a.backend == 'nccl' else
                                                                               'gloo (SMOKE MODE: not a scaling measurement)'),
                      'rccl_ranks': world if (use_dist and a.backend == 'nccl') else 0, 'gpus_visible': torch.cuda.device_count()},
            'clients': clients, 'round': rnd, 'full_M': full, 'comm': None if rnd is None else rnd['comm'], 'roofline': roof, 'cpu_baseline': cpu,
        }
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
