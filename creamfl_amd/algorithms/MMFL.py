"""Server orchestration of one CreamFL communication round.

Mirrors src/algorithms/MMFL.py:39-391: `MMFL(args, wandb)`, `set_config`, `load_dataset`, `create_model`,
`train(round_n)` (global train -> global representations -> per-client run + generate_logits -> distill ->
evaluate -> checkpoint) and `distill` with the `con_w` aggregation (row A5) and the KD steps.

What differs, by design:
  * representations never leave the GPU: global features, client representations and the con_w aggregate
    stay device tensors (the reference round-trips them through host memory and runs con_w as a
    50 000 x 50 000 fp32 matmul ON THE CPU, ~53 s and ~30 GB per client, MMFL.py:304-307).
  * with torch.distributed initialised (one process per GPU), the sampled clients of a round are sharded
    one per rank, their [M, D] representations are exchanged with ONE all-gather per client slot, and con_w is
    row-sharded (creamfl_amd/dist.py).  Without it everything runs on the single GPU, clients in sequence.
  * datasets are out of scope: `load_dataset` / `create_model` take ready loaders (any iterable with the
    reference's batch-tuple contract); by default MSCOCO-shaped synthetic loaders are built.
  * public-set size is `args.pub_data_num` everywhere (the reference hard-codes 50000 inside `aggregation`,
    MMFL.py:302,319, which only works for the default).
Reference quirks kept on purpose: the KD image term is added twice when both image and multimodal clients
exist (MMFL.py:361-378); `best_score` is never updated (MMFL.py:275-276), so "best" is every round.
"""
import gc
import operator
import os
import random

import torch
import torch.nn as nn

from .. import dist as cdist
from .. import flags, ops
from ..utils.config import default_config, parse_config
from ..utils.logger import PythonLogger
from ..utils.prefetch import DevicePrefetcher
from ..utils.synthetic import SyntheticCocoLoader
from .ClientTrainer import ClientTrainer
from .MMClientTrainer import MMClientTrainer
from .eval_coco import COCOEvaluator
from .optimizers import AdamP
from .retrieval_trainer import TrainerEngine

is_test = False


class _NullWandb:
    def log(self, *a, **k):
        pass


class MMFL(object):
    def __init__(self, args, wandb=None):
        self.args = args
        if int(flags.get(args, 'miopen_immediate') or 0):            # before the first engine is built (creamfl_amd/runtime.py)
            os.environ['CFL_MIOPEN_IMMEDIATE'] = '1'
        self.wandb = wandb if wandb is not None else _NullWandb()
        self.device = None
        self.img_local_trainers = None
        self.txt_local_trainers = None
        self.mm_local_trainers = None
        self.engine = None
        self.best_score = 0
        self.cur_epoch = 0
        self.img_train_loaders, self.txt_train_loaders = None, None
        self.dataloaders_global = None
        self.test_loader = None
        self.config = None
        self.set_config()
        self.logger = PythonLogger(output_file=None, quiet=bool(getattr(args, 'quiet', False)))
        self.img_vec, self.txt_vec = None, None
        self.global_img_feature = None
        self.global_txt_feature = None
        self.distill_index = None
        self.best_scores, self.best_metadata = None, None
        self.total_local_trainers = []
        # optional replacement of `random.sample` for the round's client choice (same signature; must return the same list on
        # every rank).  dist.balanced_sample picks one client per rank where the ownership map (client_idx % world) allows it.
        self.client_sampler = None

    def set_config(self, img='cifa100', txt='AG_NEWS'):
        """MMFL.py:70-88.  Reads ./src/coco.yaml when it exists (drop-in next to the reference tree), else
        the same values from creamfl_amd.utils.config.default_config."""
        if os.path.exists('./src/coco.yaml'):
            self.config = parse_config('./src/coco.yaml', strict_cast=False)
        else:
            self.config = default_config()
        self.config.train.model_save_path = 'model_last_no_prob.pth'
        self.config.train.best_model_save_path = 'model_best_no_prob.pth'
        self.config.train.output_file = 'model_noprob.log'
        self.config.model.img_client = img
        self.config.model.txt_client = txt
        self.config.model.embed_dim = self.args.feature_dim
        if self.args.not_bert:
            self.config.model.not_bert = True
            self.config.model.cnn_type = 'resnet50'
        else:
            self.config.model.not_bert = False
            self.config.model.cnn_type = 'resnet101'
        for k in ('cnn_type', 'bert_name'):               # build-defined encoder overrides (configs 1 and 5)
            if getattr(self.args, k, None):
                self.config.model[k] = getattr(self.args, k)
        self.config.model.wemb_type = None

    # -------------------------------------------------------------------------------------------------- data
    def _pub_key(self, eval_=False):
        return ('train_subset_eval' if eval_ else 'train_subset') + f'_{self.args.pub_data_num}'

    def load_dataset(self, args, dataloaders=None, vocab=None):
        """MMFL.py:90-114.  `dataloaders` = {'train_subset_<M>', 'train_subset_eval_<M>', 'test'}."""
        M = args.pub_data_num
        if dataloaders is None:
            bert = not self.config.model.not_bert
            bs = self.config.dataloader.batch_size
            img = getattr(args, 'image_size', 224)
            dataloaders = {
                self._pub_key(False): SyntheticCocoLoader(M, bs, seed=1, bert=bert, img=img),
                self._pub_key(True): SyntheticCocoLoader(M, 2 * bs, seed=1, bert=bert, img=img),
                'test': SyntheticCocoLoader(getattr(args, 'test_pairs', 5000), 2 * bs, seed=2, bert=bert,
                                            captions_per_image=5, img=img),
            }
        self.dataloaders_global = dataloaders
        self.vocab = vocab
        word2idx = vocab.word2idx if vocab is not None else {i: i for i in range(11755)}
        self.engine = TrainerEngine(device=self.device or 'cuda')
        self.engine.server_graph = bool(int(flags.get(args, 'server_graph')))
        self.engine.set_logger(self.logger)
        self.config.optimizer.learning_rate = self.args.server_lr
        self._dataloaders = dict(self.dataloaders_global)
        self.evaluator = COCOEvaluator(eval_method='matmul', verbose=False, eval_device=self.device or 'cuda',
                                       extract_device=self.device or 'cuda', n_crossfolds=5)
        self.engine.create(self.config, word2idx, self.evaluator, self.args.mlp_local)
        self.train_eval_dataloader = self._dataloaders.pop(self._pub_key(True), None)
        self.engine.model_to_device()
        if self.config.train.get('use_fp16'):
            self.engine.logger.log('Train with bf16 autocast (apex O2 replacement)')
            self.engine.to_half()

    def create_model(self, args, client_loaders=None, client_test_sets=None, mm_config=None):
        """MMFL.py:116-178.  client_loaders = {'img': [loader per client], 'txt': [...], 'mm': [...]}."""
        self.logger.log('start creating model and partition datasets')
        self.device = torch.device('cuda:%d' % args.device)
        client_loaders = client_loaders or {}
        client_test_sets = client_test_sets or {}
        self.img_local_trainers, self.txt_local_trainers, self.mm_local_trainers = [], [], []
        for kind, dataset, store, n in (('img', 'Cifar100', self.img_local_trainers, args.num_img_clients),
                                        ('txt', 'AG_NEWS', self.txt_local_trainers, args.num_txt_clients)):
            for i in range(n):
                t = ClientTrainer(args, dataset, None, None, None, None, self.logger,
                                  global_test_set=client_test_sets.get(kind), inter_distance=4, client_id=i,
                                  wandb=self.wandb, gpuid=str(self.device))
                loaders = client_loaders.get(kind)
                t.train_loader = loaders[i] if loaders else None
                store.append(t)
                if is_test and i == 0:
                    break
        if args.num_mm_clients > 0:
            config = mm_config
            if config is None:
                config = default_config(embed_dim=args.feature_dim, cnn_type='resnet18', not_bert=True)
                config.train.use_fp16 = False
            config.model.embed_dim = args.feature_dim
            config.model.not_bert = True
            for client_id in range(args.num_mm_clients):
                loaders = client_loaders.get('mm')
                self.mm_local_trainers.append(
                    MMClientTrainer(args, config, self.logger, client=client_id, dset_name='flicker30k',
                                    device=str(self.device), mlp_local=self.args.mlp_local,
                                    train_loader=loaders[client_id] if loaders else None))
                if is_test and client_id == 0:
                    break
        self.total_local_trainers = self.img_local_trainers + self.txt_local_trainers + self.mm_local_trainers
        for i in range(len(self.total_local_trainers)):
            self.total_local_trainers[i].client_idx = i + 1

    # -------------------------------------------------------------------------------------------------- round
    @torch.no_grad()
    def extract_global_features(self):
        """MMFL.py:194-221, device-resident: [M, D] image and caption representations of the public set."""
        loader = self.dataloaders_global[self._pub_key(True)]
        img_feature, txt_feature, distill_index = [], [], []
        eng = self.engine
        was_training = eng.model.training
        for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(DevicePrefetcher(loader, eng.device)):
            images = images.to(eng.device)
            if eng.autocast_dtype is not None:
                images = images.contiguous(memory_format=torch.channels_last)
            with torch.autocast('cuda', dtype=eng.autocast_dtype, enabled=eng.autocast_dtype is not None):
                output = eng.model(images, captions.to(eng.device), captions_word, caption_lens.to(eng.device))
            img_feature.append(output['image_features'].float())
            txt_feature.append(output['caption_features'].float())
            distill_index.extend(index)
        eng.model.train(was_training)
        ops.invalidate_bank_images()         # last round's banks (and their pre-split images) are released with the tensors below
        self.global_img_feature = torch.cat(img_feature, dim=0)
        self.global_txt_feature = torch.cat(txt_feature, dim=0)
        self.distill_index = distill_index

    def train(self, round_n):
        self.cur_epoch = round_n
        self.cur_trainers = self.total_local_trainers
        if not getattr(self, '_gc_frozen', False) and os.environ.get('CFL_NO_GC_FREEZE', '0') != '1':
            # a federation is ~27 models' worth of long-lived Python objects; every step of the host-bound loops allocates
            # thousands of short-lived ones (autograd nodes, Function contexts), so the cyclic collector keeps re-walking the
            # permanent ones.  Move what exists now out of its reach, once (objects are still freed by reference count).
            gc.collect()
            gc.freeze()
            self._gc_frozen = True
        # multi-rank: the server phases (global contrastive training, KD) are REPLICATED by default -- every rank does the whole
        # public batch: bit-for-bit the single-process round, full-batch BatchNorm statistics (the reference's semantics).
        # `--server_dp 1` (creamfl_amd/flags.py) makes them DATA-PARALLEL: every rank encodes 1/W of each public batch, features
        # are all-gathered for the full-batch loss, encoder gradients are bucket-averaged (dist.GradBuckets) -- the 391-step
        # global train and the KD loop cost 1/W instead of being repeated W times, at the price of per-shard BatchNorm batch
        # statistics (as in any data-parallel BatchNorm model); the running statistics are averaged over the ranks before the
        # evaluation and the checkpoint below.  Replicas are re-synchronised each round.
        server_dp = cdist._world()[1] > 1 and bool(int(flags.get(self.args, 'server_dp')))
        if server_dp:
            if self.engine.dp is None:
                self.engine.enable_data_parallel(bucket_cap_mb=int(flags.get(self.args, 'bucket_mb')))
            self.engine.shard_batches = True
        elif self.engine.dp is not None:
            self.engine.shard_batches = False
        self.engine.sync_replicas()
        if not is_test:
            self.logger.log(f"Round {round_n + 1}!")
            self.engine.train(tr_loader=self._dataloaders[self._pub_key(False)])
            if len(self.total_local_trainers) != 0:
                # every rank must sample the same clients: the python RNG is seeded identically (main.py)
                sampler = getattr(self, 'client_sampler', None) or random.sample      # (MMFL.py:223: random.sample)
                self.cur_trainers = sampler(self.total_local_trainers, self.args.client_num_per_round)
        if self.args.agg_method == "con_w" or self.args.contrast_local_intra or self.args.contrast_local_inter:
            self.extract_global_features()

        rank, world = cdist._world()
        my_trainers = cdist.shard_clients(self.cur_trainers, rank, world)
        gather = None
        if world > 1:
            # (8f-3) one pre-allocated [W, K, M, D] buffer per run: the clients write their representations into this rank's
            # slices, ONE all-gather moves them (optionally as bf16: --rep_wire bf16)
            M, D = self.args.pub_data_num, self.args.feature_dim
            wire = torch.bfloat16 if flags.get(self.args, 'rep_wire') == 'bf16' else torch.float32
            plan = cdist.client_plan(self.cur_trainers, world)
            gather = getattr(self, '_rep_gather', None)
            if gather is None or not gather.matches(plan, M, D, wire):
                gather = self._rep_gather = cdist.RepGatherBuffer(plan, M, D, self.engine.device, wire)
            else:
                gather.rebind(plan)
        local_reps = []
        for slot, trainer in enumerate(my_trainers):
            self.logger.log(f"Training Client {trainer.client_idx}!")
            trainer.cur_epoch = round_n
            trainer.run(self.global_img_feature, self.global_txt_feature, self.distill_index,
                        self._dataloaders[self._pub_key(False)])
            self.logger.log("Generate Local Representations")
            if gather is not None:
                _vec, i = trainer.generate_logits(self.dataloaders_global[self._pub_key(True)], out=gather.out_views(slot))
            else:
                _vec, i = trainer.generate_logits(self.dataloaders_global[self._pub_key(True)])
            if self.distill_index is None:
                self.distill_index = i
            else:
                assert i == self.distill_index
            local_reps.append(_vec)
        if gather is not None:
            gather.gather()
            img_vec, txt_vec = gather.blocks()
        else:
            img_vec = [v['img'] for v in local_reps if v['img'] is not None]
            txt_vec = [v['txt'] for v in local_reps if v['txt'] is not None]

        if not self.args.disable_distill:
            self.distill(round_n, img_vec, txt_vec, None, None, self.distill_index)

        if server_dp:
            # per-shard BatchNorm statistics: every rank evaluates (and rank 0 saves) the SAME model -- the mean of the ranks'
            # running statistics, not rank 0's shard-local ones
            self.engine.average_running_stats()
        metadata = self.engine.metadata.copy()
        metadata['cur_epoch'] = round_n + 1
        metadata['lr'] = self.engine.optimizer.param_groups[0]['lr']
        test_ds = self._dataloaders['test'].dataset
        fold_kw = {}
        if test_ds.n_images < 5000:          # smaller-than-COCO-5K test sets: 5 equal folds of what there is
            fold_kw = dict(n_images_per_crossfold=test_ds.n_images // 5, n_captions_per_crossfold=len(test_ds) // 5)
        test_scores = self.engine.evaluate({'test': self._dataloaders['test']}, **fold_kw)
        self.engine.report_scores(step=round_n + 1, scores=test_scores, metadata=metadata,
                                  prefix=self.engine.eval_prefix)
        t = test_scores['test']
        rsum = t['i2t']['recall_1'] + t['t2i']['recall_1']
        if 'n_fold' in t:
            rsum += t['n_fold']['i2t']['recall_1'] + t['n_fold']['t2i']['recall_1']
            self.wandb.log({"Server n_fold_i2t_r1": t['n_fold']['i2t']['recall_1']}, step=self.cur_epoch)
            self.wandb.log({"Server n_fold_t2i_r1": t['n_fold']['t2i']['recall_1']}, step=self.cur_epoch)
        self.wandb.log({"Server rsum_r1": rsum}, step=self.cur_epoch)
        self.wandb.log({"Server i2t_r1": t['i2t']['recall_1']}, step=self.cur_epoch)
        self.wandb.log({"Server t2i_r1": t['t2i']['recall_1']}, step=self.cur_epoch)
        if self.best_score < rsum:
            metadata['best_score'] = rsum                 # (the reference never stores it back: MMFL.py:275-276)
            metadata['best_epoch'] = round_n + 1
            self.best_metadata, self.best_scores = metadata, test_scores
            if rank == 0 and getattr(self.args, 'save_checkpoints', True):
                torch.save({'net': self.engine.model_state_dict()}, self.args.name + '-best_model.pt')
        if round_n == self.args.comm_rounds - 1 and rank == 0 and getattr(self.args, 'save_checkpoints', True):
            torch.save({'net': self.engine.model_state_dict()}, self.args.name + '-last_model.pt')
        self.engine.lr_scheduler.step()
        del img_vec, txt_vec
        gc.collect()

    # -------------------------------------------------------------------------------------------------- distill
    def aggregation(self, i_vec, t_vec):
        """MMFL.py:298-335 (`con_w`): log-prob of every client representation against the other modality's global
        bank, softmax over clients, weighted sum -- on the GPU (csrc/bank.hip), row-sharded across ranks."""
        if self.args.agg_method != "con_w":
            raise NotImplementedError
        if i_vec:
            i_vec = cdist.conw_aggregate_sharded(i_vec, self.global_txt_feature)
        if t_vec:
            t_vec = cdist.conw_aggregate_sharded(t_vec, self.global_img_feature)
        return i_vec, t_vec

    def kd_terms(self, output, d_idx):
        """The KD loss of one public batch (MMFL.py:355-378): one `kd_weight * MSELoss(output, agg[d_idx])` per client
        type, each a fused gather + MSE kernel (ops.kd_mse).  As in the reference the image term appears once in the
        image-client block and once more in the multimodal block, so it counts twice when both kinds of client exist.
        Returns 0 (python int) when no term applies."""
        def code_sim(output, agg):
            output = output.sum(axis=1) if len(output.shape) == 3 else output
            return ops.kd_mse(output, agg, d_idx, self.args.kd_weight)

        has_img = self.img_vec is not None and len(self.img_vec)
        has_txt = self.txt_vec is not None and len(self.txt_vec)
        loss = 0
        if self.args.num_img_clients > 0 and has_img:
            loss = loss + code_sim(output['image_features'], self.img_vec)
        if self.args.num_txt_clients > 0 and has_txt:
            loss = loss + code_sim(output['caption_features'], self.txt_vec)
        if self.args.num_mm_clients > 0:
            if has_img:
                loss = loss + code_sim(output['image_features'], self.img_vec)
            if has_txt:
                loss = loss + code_sim(output['caption_features'], self.txt_vec)
        return loss

    def _kd_has_terms(self):
        has_img = self.img_vec is not None and len(self.img_vec)
        has_txt = self.txt_vec is not None and len(self.txt_vec)
        a = self.args
        return bool(((a.num_img_clients > 0 or a.num_mm_clients > 0) and has_img)
                    or ((a.num_txt_clients > 0 or a.num_mm_clients > 0) and has_txt))

    def _kd_graph_fn(self, model):
        eng = self.engine

        def fn(images, captions, caption_lens, d_idx):
            ops.dropout_tick(images.device).add_(1)
            with ops.join_scope():
                with torch.autocast('cuda', dtype=eng.autocast_dtype, enabled=eng.autocast_dtype is not None):
                    output = model(images, captions, None, caption_lens)
                loss = self.kd_terms(output, d_idx)
                eng.backward_and_step(loss)
            return loss.detach()
        return fn

    def distill(self, round_n, img_vec, txt_vec, img_num, txt_num, distill_index):
        # Multi-rank: the server phases are replicated, and the replicas stay identical only if their dropout draws are --
        # but the ranks have just trained DIFFERENT clients and consumed the generators differently.  Rank 0 draws a seed,
        # every rank re-seeds with it.
        cdist.reseed_from_rank0(self.engine.device)
        self.engine.model.train()
        img_vec, txt_vec = self.aggregation(img_vec, txt_vec)
        self.img_vec, self.txt_vec = img_vec, txt_vec
        distill_dict = {b: a for a, b in enumerate(distill_index)}
        self.logger.log("start distilling")
        eng = self.engine
        model = eng.dp.module if eng.dp is not None else eng.model

        for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(
                DevicePrefetcher(self.dataloaders_global[self._pub_key(False)], eng.device)):
            images = images.to(eng.device)
            captions, caption_lens = captions.to(eng.device), caption_lens.to(eng.device)
            sh = eng.batch_shard(images.shape[0])
            if sh is not None:        # data-parallel KD: mean-MSE over this rank's rows; the bucket average over ranks = full mean
                r0, r1 = sh
                images, captions, caption_lens, index = images[r0:r1], captions[r0:r1], caption_lens[r0:r1], index[r0:r1]
                captions_word = captions_word[r0:r1] if captions_word is not None else None
            if eng.autocast_dtype is not None:
                images = images.contiguous(memory_format=torch.channels_last)
            d_idx = operator.itemgetter(*index)(distill_dict)
            d_idx = torch.as_tensor(d_idx if isinstance(d_idx, tuple) else (d_idx,), device=eng.device)
            if sh is None and self._kd_has_terms() and eng.graph_capable(captions_word):
                # --server_graph 1: the KD step from a HIP graph, one capture per round (the aggregated representations and the
                # learning rate are constants of a round)
                key = (id(eng.model), tuple(g['lr'] for g in eng.optimizer.param_groups),
                       tuple(v.data_ptr() if torch.is_tensor(v) else None for v in (self.img_vec, self.txt_vec)))
                gs = eng.graphed_step('kd', key, self._kd_graph_fn(model))
                gs(*eng.graph_inputs(gs, images, captions, caption_lens), d_idx, device=images.device)
                continue
            with ops.join_scope():                # the same backward path as the contrastive step (gradient joins fused)
                with torch.autocast('cuda', dtype=eng.autocast_dtype, enabled=eng.autocast_dtype is not None):
                    output = model(images, captions, captions_word, caption_lens)
                loss = self.kd_terms(output, d_idx)
                if not torch.is_tensor(loss):
                    continue
                eng.backward_and_step(loss)       # incl. the bucketed gradient averaging when data parallel is on
            del output, loss                      # (no live autograd graph across iterations: see MMClientTrainer._local_epoch)
        eng.drop_graph('kd')
