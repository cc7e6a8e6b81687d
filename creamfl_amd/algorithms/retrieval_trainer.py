"""Server engine (row S1: the contrastive step the headline metric counts).

Mirrors src/algorithms/retrieval_trainer.py:37-237 (EngineBase / TrainerEngine): create,
model_to_device, to_half, train, evaluate, save_models / load_models, report_scores, and the attributes
main.py / MMFL.py read (.model .optimizer .criterion .lr_scheduler .device .metadata .logger .eval_prefix).

MI355X specifics:
  * `to_half()` replaces apex amp O2 (:107-111) by bf16 autocast of the encoder trunks + channels_last
    convolutions; the contrastive head and loss stay fp32 (apex also hands fp32 outputs to the criterion).
  * `train_step()` is the per-batch body of `train()` (:192-214): forward -> MCSoftContrastiveLoss (HIP) ->
    zero_grad -> backward -> clip_grad_norm_(model.parameters(), grad_clip) -> AdamP.step, with no host
    synchronisation inside the step.
  * with torch.distributed initialised (one process per GPU, RCCL), `enable_data_parallel()` turns the step
    into large-batch global contrast: per-rank features are all-gathered (creamfl_amd/dist.py) and the encoder
    gradients are summed by bucketed all-reduce overlapped with backward.
"""
import contextlib
import hashlib
import json

import torch
import torch.nn as nn

from .. import runtime
from ..criterions import get_criterion
from ..networks.models import get_model
from ..utils.prefetch import DevicePrefetcher
from ..utils.serialize_utils import flatten_dict
from .optimizers import AdamP, get_lr_scheduler, get_optimizer


def get_lr(optimizer):
    for param_group in optimizer.param_groups:
        return param_group['lr']


class EngineBase(object):
    def __init__(self, device='cuda', partition_train_distill=-1.):
        self.device = device
        self.model = None
        self.optimizer = None
        self.criterion = None
        self.lr_scheduler = None
        self.evaluator = None
        self.config = None
        self.logger = None
        self.metadata = {}
        self.partition_train_distill = partition_train_distill
        self.autocast_dtype = None
        self.dp = None                       # creamfl_amd.dist.DataParallelContext when enabled
        self.shard_batches = False           # multi-rank: train() / the KD loop give every rank 1/W of each batch (MMFL --server_dp)
        self._conv1x1_weights = None         # weights whose transposes are prepared in one launch before backward
        self._clip_params = None             # (id(model), its parameter list) for the gradient clip
        self.server_graph = False            # --server_graph: the contrastive / KD steps replayed from HIP graphs (graphed_step)
        self._graphs = {}                    # name -> (key, GraphedStep)
        self.graph_stats = {}                # name -> {'calls', 'replays', 'failed'} of the last graph of that name

    def create(self, config, word2idx, evaluator, mlp_local):
        runtime.configure()                  # MIOpen find mode / recorded find-db / cudnn.benchmark before the first convolution
        self.config = config
        self.word2idx = word2idx
        self.model = get_model(word2idx, config.model, mlp_local)
        self.set_criterion(get_criterion(config.criterion.name, config.criterion))
        params = [param for param in self.model.parameters() if param.requires_grad]
        params += [param for param in self.criterion.parameters() if param.requires_grad]
        self.set_optimizer(get_optimizer(config.optimizer.name, params, config.optimizer))
        self.set_lr_scheduler(get_lr_scheduler(config.lr_scheduler.name, self.optimizer, config.lr_scheduler))
        if evaluator is not None:
            evaluator.set_model(self.model)
            evaluator.set_criterion(self.criterion)
            self.set_evaluator(evaluator)
        if self.logger is not None:
            self.logger.log('Engine is created.')
            self.logger.update_tracker({'full_config': dict(config)}, keys=['full_config'])
        self.prefix = 'train__'
        self.eval_prefix = ''
        if self.logger is not None:
            self.logger.log('start train')
        self.img_code, self.txt_code, self.mm_txt_code, self.mm_img_code = None, None, None, None

    def model_to_device(self):
        self.model.to(self.device)
        if self.criterion:
            self.criterion.to(self.device)

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer

    def set_criterion(self, criterion):
        self.criterion = criterion

    def set_lr_scheduler(self, lr_scheduler):
        self.lr_scheduler = lr_scheduler

    def set_evaluator(self, evaluator):
        self.evaluator = evaluator
        self.evaluator.set_logger(self.logger)

    def set_logger(self, logger):
        self.logger = logger

    def to_half(self, bf16_weights=True):
        """Mixed precision without apex (the reference: amp.initialize(opt_level='O2'), :107-111): bf16 autocast for
        the encoder trunks, channels_last convolutions and -- like O2 -- low-precision trunk WEIGHTS with fp32
        master copies owned by the optimizer (creamfl_amd AdamP only): the per-step autocast weight casts and the
        bf16->fp32 gradient casts disappear and the data-parallel gradient all-reduce halves.  BatchNorm /
        LayerNorm parameters, the PIE head, the projection heads and the criterion stay fp32."""
        self.autocast_dtype = torch.bfloat16
        self.model.to(memory_format=torch.channels_last)
        if bf16_weights and isinstance(self.optimizer, AdamP):
            trunks = [getattr(self.model, 'img_enc', None) and self.model.img_enc.cnn]
            if not self.model.config.not_bert:
                trunks.append(self.model.txt_enc)
            for trunk in trunks:
                if trunk is None:
                    continue
                for mod in trunk.modules():
                    if isinstance(mod, (nn.Conv2d, nn.Linear, nn.Embedding)):
                        for p in mod.parameters(recurse=False):
                            if p.dtype == torch.float32 and p.requires_grad:
                                self.optimizer.make_master(p)
                                p.data = p.data.to(torch.bfloat16)
        if self.evaluator is not None:
            self.evaluator.autocast_dtype = self.autocast_dtype

    def sync_replicas(self, src=0, group=None):
        """Multi-rank: make this rank's server replica identical to rank `src`'s -- parameters and buffers (with the
        BatchNorm batch counters flushed first), and the optimizer state: with bf16 trunk weights the next step
        rewrites every weight from the rank's own fp32 master, so the masters and moments must travel too."""
        from .. import dist as cdist
        if cdist._world(group)[1] == 1:
            return
        for m in self.model.modules():
            if hasattr(m, 'flush_num_batches_tracked'):
                m.flush_num_batches_tracked()
        cdist.broadcast_module(self.model, src, group)
        cdist.broadcast_module(self.criterion, src, group)
        if isinstance(self.optimizer, AdamP):
            self.optimizer.broadcast_state(src, group)

    def average_running_stats(self, group=None):
        """Data-parallel server phases: BatchNorm batch statistics are per shard, so the running statistics drift apart
        between the round-start syncs.  Average them over the ranks (one flat all-reduce) so that every rank evaluates, and
        rank 0 checkpoints, the same model."""
        from .. import dist as cdist
        if cdist._world(group)[1] == 1:
            return
        for m in self.model.modules():
            if hasattr(m, 'flush_num_batches_tracked'):
                m.flush_num_batches_tracked()
        cdist.average_buffers(self.model, group)

    def enable_data_parallel(self, process_group=None, bucket_cap_mb=32):
        from ..dist import DataParallelContext
        fused = isinstance(self.optimizer, AdamP)
        if self.dp is not None:
            self.dp.close()                      # a second reducer on the same parameters would reduce everything twice
        self.dp = DataParallelContext(self.model, process_group, bucket_cap_mb=bucket_cap_mb, assign_grads=not fused)
        if fused:
            # the fused optimizer reads the averaged gradients straight from the bucket views (no per-parameter grad
            # re-assignment on the host); `consume` makes it refuse views that no finish_backward() has filled
            self.optimizer.grad_override = self.dp.reducer.grad_views()
            self.optimizer.grad_override_consume = self.dp.reducer.consume
        # Replicas must agree on more than module state: with bf16 trunk weights the next step rewrites every weight from
        # the rank's own fp32 master, so masters, moments and step counts travel too (and the criterion's scalars).
        self.sync_replicas(group=process_group)

    def disable_data_parallel(self):
        if self.dp is not None:
            self.dp.close()
            self.dp = None
        if isinstance(self.optimizer, AdamP):
            self.optimizer.grad_override = None
            self.optimizer.grad_override_consume = None

    @torch.no_grad()
    def evaluate(self, val_loaders, n_crossfolds=None, **kwargs):
        if self.evaluator is None:
            if self.logger is not None:
                self.logger.log('[Evaluate] Warning, no evaluator is defined. Skip evaluation')
            return
        self.model_to_device()
        self.model.eval()
        if not isinstance(val_loaders, dict):
            val_loaders = {'te': val_loaders}
        scores = {}
        for key, data_loader in val_loaders.items():
            if 'train' in key:
                continue
            if self.logger is not None:
                self.logger.log('Evaluating {}...'.format(key))
            _n_crossfolds = -1 if key == 'val' else n_crossfolds
            scores[key] = self.evaluator.evaluate(data_loader, n_crossfolds=_n_crossfolds, key=key, **kwargs)
        return scores

    def save_models(self, save_to, metadata=None):
        state_dict = {
            'model': self.model_state_dict(), 'criterion': self.criterion.state_dict(),
            'optimizer': self.optimizer.state_dict(), 'lr_scheduler': self.lr_scheduler.state_dict(),
            'config': json.loads(json.dumps(self.config, default=str)), 'word2idx': self.word2idx, 'metadata': metadata,
        }
        torch.save(state_dict, save_to)
        if self.logger is not None:
            self.logger.log('state dict is saved to {}, metadata: {}'.format(save_to, json.dumps(metadata, indent=4)))

    def model_state_dict(self):
        """The model's state dict with full-precision weights: after to_half() the trunk weights the model computes
        with are bf16 and their fp32 masters live in the optimizer -- checkpoints store the masters (what the
        reference's apex-O2 checkpoints hold)."""
        if isinstance(self.optimizer, AdamP):
            return self.optimizer.master_state_dict(self.model)
        return self.model.state_dict()

    def load_model_weights(self, model_state, strict=True):
        """model.load_state_dict + re-derivation of the optimizer's fp32 masters from the loaded values (fp32 values
        of the checkpoint where it has them), so that the next step does not restore stale weights."""
        res = self.model.load_state_dict(model_state, strict=strict)
        if isinstance(self.optimizer, AdamP):
            named = dict(self.model.named_parameters())
            fp32 = {named[k]: v for k, v in model_state.items()
                    if k in named and torch.is_tensor(v) and v.dtype == torch.float32 and named[k].dtype == torch.bfloat16}
            self.optimizer.refresh_masters(fp32)
        return res

    def load_models(self, state_dict_path, load_keys=None):
        with open(state_dict_path, 'rb') as fin:
            self.metadata['pretrain_hash'] = hashlib.sha1(fin.read()).hexdigest()
        state_dict = torch.load(state_dict_path, map_location='cpu')
        if 'model' not in state_dict:
            self.load_model_weights(state_dict.get('net', state_dict), strict=False)
            return
        # optimizer BEFORE model: the optimizer state carries the masters of the step the checkpoint was taken at; loading
        # the model afterwards re-derives them from the (fp32) checkpoint weights, which are the same values
        keys = list(load_keys or ['model', 'criterion', 'optimizer', 'lr_scheduler'])
        for key in sorted(keys, key=lambda k: k == 'model'):
            if key == 'model':
                try:
                    self.load_model_weights(state_dict[key])
                except RuntimeError as e:
                    if self.logger is not None:
                        self.logger.log('Unable to import state_dict, missing keys are found. {}'.format(e))
                    self.load_model_weights(state_dict[key], strict=False)
                continue
            try:
                getattr(self, key).load_state_dict(state_dict[key])
            except RuntimeError as e:
                if self.logger is not None:
                    self.logger.log('Unable to import state_dict, missing keys are found. {}'.format(e))
                getattr(self, key).load_state_dict(state_dict[key], strict=False)


class TrainerEngine(EngineBase):

    def batch_shard(self, n):
        """(r0, r1): this rank's rows of an n-row server batch when the server phases run data-parallel, or None when the
        batch stays whole on every rank (single process, sharding off, or n not divisible by the world size -- the
        collectives move equal blocks; such a batch is simply processed replicated, which gives the same update)."""
        if self.dp is None or not self.shard_batches or self.dp.world == 1 or n % self.dp.world:
            return None
        per = n // self.dp.world
        return self.dp.rank * per, (self.dp.rank + 1) * per

    # ---- the server's steps from HIP graphs (--server_graph 1) ---------------------------------------------------------------
    # At the reference's public batch (128) the server step is host-bound: ~1 600 launches issued in 27-35 ms of Python for ~24 ms
    # of GPU work, 782 times per round (global training + KD).  A step is capturable once nothing in it travels as a host value:
    # the fused AdamP's step count (cfl_adamp_step_counted), the fused BERT dropout's seeds (ops.dropout_tick), the BatchNorm batch
    # counters (on the device since the clients' graphs), captions padded to ONE width (the attention mask comes from the lengths
    # on the device).  One capture per phase and round: the learning rate is a constant of a round (MMFL.py:286), and the KD step
    # skips the criterion's scalars, which voids the contrastive step's captured step-count offsets (AdamP.CaptureHandle).
    def graph_capable(self, captions_word=None):
        return bool(self.server_graph and self.dp is None and isinstance(self.optimizer, AdamP)
                    and torch.device(self.device).type == 'cuda' and torch.cuda.is_available()
                    and (captions_word is None or getattr(self.model, 'tokenizer', None) is None))

    def graphed_step(self, name, key, fn):
        """The GraphedStep of phase `name`, re-made (the old one dropped first: its pool and pinned tables return) when `key`
        changes."""
        from ..graphs import GraphedStep
        slot = self._graphs.get(name)
        if slot is None or slot[0] != key:
            self.drop_graph(name)
            log = self.logger.log if self.logger is not None else None
            gs = GraphedStep(fn, warmup=3, log=log, optimizer=self.optimizer, other_threads=True)
            gs.caption_width = None
            self._graphs[name] = (key, gs)
        return self._graphs[name][1]

    def drop_graph(self, name):
        slot = self._graphs.pop(name, None)
        if slot is not None:
            gs = slot[1]
            self.graph_stats[name] = {'calls': gs.calls, 'replays': gs.replays, 'failed': gs.failed}

    @staticmethod
    def graph_caption_width(first_width):
        """Padded caption width of a captured server step: the first batch's, rounded up to a multiple of 8 (a public batch of 128
        captions nearly always contains one of the maximum length); a wider batch runs eagerly."""
        return (int(first_width) + 7) // 8 * 8

    def graph_inputs(self, gs, images, captions, caption_lens):
        """(images, captions, lengths) in the form the captured step takes them: channels_last images when the trunk computes in
        that layout, captions zero-padded to the graph's width, int64 lengths."""
        from .ClientTrainer import pad_captions
        if gs.caption_width is None:
            gs.caption_width = self.graph_caption_width(captions.shape[1])
        if self.autocast_dtype is not None and images.dim() == 4:
            images = images.contiguous(memory_format=torch.channels_last)
        return images, pad_captions(captions, gs.caption_width), caption_lens.to(torch.int64)

    def _train_graph_fn(self):
        from .. import ops

        def fn(images, captions, caption_lens):
            ops.dropout_tick(images.device).add_(1)
            loss, _ = self.train_step(images, captions, None, caption_lens, gather=False)
            return loss.detach()
        return fn

    def backward_and_step(self, loss):
        """zero_grad -> backward -> (multi-rank: bucketed gradient averaging) -> clip -> optimizer step: the tail every
        server-side step shares (the contrastive step, retrieval_trainer.py:208-214, and the KD step, MMFL.py:385-391)."""
        self.optimizer.zero_grad(set_to_none=True)
        if self.dp is not None:
            self.dp.prepare_backward()
        self.backward(loss)
        if self.dp is not None:
            self.dp.finish_backward(list(self.criterion.parameters()))
        self.optimizer_step()

    def forward_loss(self, images, captions, captions_word, caption_lens, gather=True):
        model = self.dp.module if self.dp is not None else self.model
        with torch.autocast('cuda', dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            output = model(images, captions, captions_word, caption_lens)
        if self.dp is not None and gather:
            output = dict(output)
            output['image_features'], output['caption_features'] = self.dp.gather_features(
                output['image_features'], output['caption_features'])
        loss, loss_dict = self.criterion(**output)
        return loss, loss_dict

    def train_step(self, images, captions, captions_word, caption_lens, gather=True):
        """One server contrastive step (retrieval_trainer.py:192-214).  With data parallel on, `gather=True` means the
        arguments are THIS RANK'S part of the global batch (features are all-gathered, bench.py / sharded server phases);
        `gather=False` means every rank holds the whole batch (replicated step: the averaged gradients are the gradients)."""
        if self.autocast_dtype is not None and images.dim() == 4:
            images = images.contiguous(memory_format=torch.channels_last)
        from .. import ops
        # the backward of this forward runs through self.backward(): the gradient joins of the residual blocks may fuse into the
        # data-gradient GEMMs (ops.JOIN), armed for exactly this step
        with (ops.join_scope() if images.is_cuda else contextlib.nullcontext()):
            loss, loss_dict = self.forward_loss(images, captions, captions_word, caption_lens, gather=gather)
            self.backward_and_step(loss)
        return loss, loss_dict

    def backward(self, loss):
        """loss.backward() with the weight transforms of the trunk's data gradients (W^T of the 1x1 convolutions for the GEMM,
        the rotated k x k weights for the forward-kernel data gradients) written by ONE launch beforehand."""
        if not loss.is_cuda:
            loss.backward()
            return
        from .. import ops
        if self._conv1x1_weights is None:
            self._conv1x1_weights = [m.weight for m in self.model.modules()
                                     if isinstance(m, nn.Conv2d) and m.stride == (1, 1) and m.groups == 1
                                     and (m.kernel_size == (1, 1) or m.padding == (m.kernel_size[0] // 2,) * 2)]
        ops.prepare_weight_transposes(self._conv1x1_weights)
        try:
            with runtime.backward_here():            # (no hand-over to autograd's worker thread: the step is host-bound at batch 128)
                loss.backward()
        finally:
            ops.release_weight_transposes()

    def optimizer_step(self):
        """clip_grad_norm_(model.parameters(), grad_clip) + optimizer.step() (:211-214); fused into the
        multi-tensor HIP kernels when the optimizer is creamfl_amd's AdamP."""
        clip = self.config.train.grad_clip
        if clip > 0:
            # the module tree is walked once, not every step (500 modules: 1.3 ms of a 29 ms host-bound step); a model swapped in
            # later (load_models / set_model) shows up as another id
            if getattr(self, '_clip_params', None) is None or self._clip_params[0] != id(self.model):
                self._clip_params = (id(self.model), list(self.model.parameters()))
        if isinstance(self.optimizer, AdamP):
            self.optimizer.step(clip=(self._clip_params[1], clip) if clip > 0 else None)
        else:
            if clip > 0:
                nn.utils.clip_grad.clip_grad_norm_(self._clip_params[1], clip)
            self.optimizer.step()

    def train(self, tr_loader, pub_data_ratio=1.):
        self.model.train()
        if self.logger is not None:
            self.logger.log("Global Training!")
        n_batches = len(tr_loader)
        # batch k+1 is pinned and copied on a side stream while batch k trains (utils/prefetch.py); the reference copies at the
        # top of every iteration (retrieval_trainer.py:194-196)
        for idx, (images, captions, captions_word, caption_lens, a_, b_, index) in enumerate(DevicePrefetcher(tr_loader, self.device)):
            images = images.to(self.device, non_blocking=True)
            captions = captions.to(self.device, non_blocking=True)
            caption_lens = caption_lens.to(self.device, non_blocking=True)
            if idx == int(n_batches * pub_data_ratio):
                break
            sh = self.batch_shard(images.shape[0])
            if sh is not None:          # multi-rank: 1/W of the batch per rank, features all-gathered, gradients bucket-averaged
                r0, r1 = sh
                cw = captions_word[r0:r1] if captions_word is not None else None
                self.train_step(images[r0:r1], captions[r0:r1], cw, caption_lens[r0:r1])
            elif self.graph_capable(captions_word):
                gs = self.graphed_step('train', (id(self.model), tuple(g['lr'] for g in self.optimizer.param_groups)),
                                       self._train_graph_fn())
                gs(*self.graph_inputs(gs, images, captions, caption_lens), device=images.device)
            else:
                self.train_step(images, captions, captions_word, caption_lens, gather=False)
        self.drop_graph('train')        # (the KD steps that follow void its step-count offsets; its activations' pool returns)

    def report_scores(self, step, scores, metadata, prefix=''):
        report_dict = {data_key: flatten_dict(_scores, sep='_') for data_key, _scores in scores.items()}
        report_dict = flatten_dict(report_dict, sep='__')
        tracker_data = report_dict.copy()
        report_dict = {'{}{}'.format(prefix, key): val for key, val in report_dict.items()}
        report_dict['step'] = step
        if 'lr' in metadata:
            report_dict['{}lr'.format(prefix)] = metadata['lr']
        keys = ['n_fold_i2t_recall_1', 'n_fold_i2t_recall_5', 'n_fold_i2t_recall_10', 'n_fold_t2i_recall_1',
                'n_fold_t2i_recall_5', 'n_fold_t2i_recall_10', 'i2t_recall_1', 'i2t_recall_5', 'i2t_recall_10',
                't2i_recall_1', 't2i_recall_5', 't2i_recall_10']
        report_dict['summary'] = ', '.join(str(report_dict.get(f'{prefix}test__{k}')) for k in keys)
        if self.logger is not None:
            self.logger.report(report_dict, prefix='[Eval] Report @step: ', pretty=True)
        tracker_data['metadata'] = metadata
        tracker_data['scores'] = scores
        if self.logger is not None:
            self.logger.update_tracker(tracker_data)
