"""Uni-modal client (rows A2c + A3 + A4 in their caller).  Mirrors src/algorithms/ClientTrainer.py:136-674:
same constructor signature, `.train_loader` assigned by the owner (MMFL.py:136), `.client_idx`, `.cur_epoch`,
`run(global_img_feature, global_txt_feature, distill_index, global_train_loader)`, `generate_logits(dataloader)`.

Out of scope here: the CIFAR-100 / AG_NEWS dataset classes and transforms -- loaders are
supplied by the caller (`train_loader`, `global_test_set`); only their batch contracts are used:
  image client  : (inputs [B,3,H,W] f32, labels [B] i64)
  text client   : (token ids [B,L] i64, labels [B] i64, lengths [B] i64)
Representations stay on the GPU between server and clients (the reference bounces them through host memory,
MMFL.py:209-210, ClientTrainer.py:370,651); `generate_logits` therefore returns device tensors.
"""
import copy
import operator
import os

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import flags, losses, ops, runtime
from ..graphs import GraphedStep
from ..networks.language_model import EncoderText
from ..networks.resnet_client import resnet18_client
from ..utils.Utils import to_one_hot
from .contrast import client_contrast_loss

is_test = False


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """Computes the precision@k for the specified values of k (ClientTrainer.py:113-129)."""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred).to(pred.device))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / batch_size) for k in topk]


IMAGE_SETS = ('Cifar100', 'Cifar10')
TEXT_SETS = ('AG_NEWS', 'YelpReviewPolarity')
CLASSES = {'Cifar100': 100, 'Cifar10': 10, 'AG_NEWS': 4, 'YelpReviewPolarity': 2}


def caption_graph_width(first_width):
    """Padded caption width of a text client's captured contrast step: the first batch's width plus headroom, a multiple of 8, at
    least 32 (COCO captions: the batch maximum is rarely above 30 words).  Wider batches run eagerly."""
    return max(32, (int(first_width) + 8 + 7) // 8 * 8)


def pad_captions(captions, width):
    """[B, L] token ids -> [B, width] with the pad token 0 (the loaders' own padding); unchanged when already that wide or wider."""
    if captions.shape[1] >= width:
        return captions
    return torch.nn.functional.pad(captions, (0, width - captions.shape[1]))


class ClientTrainer:
    def __init__(self, args, dataset, dst, RGBmean, RGBstdv, data_dict, logger, global_test_set, inter_distance=4,
                 loss='softmax', gpuid='cuda:0', num_epochs=30, init_lr=0.0001, decay=0.1, batch_size=512,
                 imgsize=256, num_workers=4, print_freq=10, save_step=10, scale=128, pool_type='max_avg',
                 client_id=-1, wandb=None):
        runtime.configure()                  # same library set-up as the server engine (creamfl_amd/runtime.py)
        torch.manual_seed(0)
        self.args = args
        if dataset == 'Flickr30k':
            init_lr = 0.0002
        self.client_id = client_id
        self.dset_name = dataset
        self.dst = dst
        self.gpuid = gpuid
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.decay_time = [False, False]
        self.init_lr = init_lr
        self.decay_rate = decay
        self.num_epochs = num_epochs
        self.cur_epoch = -1
        self.data_dict = data_dict
        self.imgsize = imgsize
        self.RGBmean, self.RGBstdv = RGBmean, RGBstdv
        self.record = []
        self.epoch = 0
        self.print_freq = print_freq
        self.save_step = save_step
        self.loss = loss
        self.losses = AverageMeter()
        self.top1, self.test_top1 = AverageMeter(), AverageMeter()
        self.top5, self.test_top5 = AverageMeter(), AverageMeter()
        self.scale = scale
        self.pool_type = pool_type
        self.inter_distance = inter_distance
        if dst and not os.path.exists(dst):
            os.makedirs(dst, exist_ok=True)
        self.logger = logger
        self.wandb = wandb
        self.loadData()
        self.setModel()
        self.old_model = None
        self.local_epochs = args.local_epochs
        self.local_epoch = 0
        self.global_test_set = global_test_set
        self.train_loader = None
        self.client_idx = -1
        # image encoders: channels_last by default (same fp32 arithmetic, the library's NHWC kernels), bf16 autocast as an opt-in
        is_cuda = torch.device(gpuid).type == 'cuda'
        self._cl = bool(self.dset_name in IMAGE_SETS and is_cuda and int(flags.get(args, 'client_channels_last')))
        self._bf16 = bool(self.dset_name in IMAGE_SETS and is_cuda and int(flags.get(args, 'client_bf16')))
        if self._bf16:
            self._cl = True
        if is_cuda and int(flags.get(args, 'client_conv_x3')):
            from .. import ops
            ops.X3CONV[0] = True                 # process-wide: every fp32 channels_last 3 x 3 / stride-1 convolution (flags.py)

    def _log(self, msg):
        if self.logger is not None:
            self.logger.log(msg)

    # -- step 1: data (class counts only; loaders are injected) ---------------------------------------------
    def loadData(self):
        if self.dset_name not in CLASSES:
            assert False, 'Dataset Not Supported!'
        self.classSize = CLASSES[self.dset_name]
        self.class_label_coco = torch.Tensor(np.array(range(80)))
        self.class_label = torch.Tensor(np.array(range(self.classSize)))

    # -- step 2: model ------------------------------------------------------------------------------------------
    def setModel(self):
        self._log(f'Setting model {self.client_id}')
        if self.dset_name in IMAGE_SETS:
            self.model = resnet18_client(pretrained=True, num_class=self.classSize, pool_type=self.pool_type,
                                         is_train=True, scale=self.scale, mlp_local=self.args.mlp_local,
                                         embed_dim=self.args.feature_dim)
        else:
            self.model = EncoderText(embed_dim=self.args.feature_dim, num_class=self.classSize, scale=self.scale,
                                     mlp_local=self.args.mlp_local)
        self.criterion = losses.create(self.loss)
        self.center_criterion = nn.MSELoss()
        self.optimizer = optim.SGD(self.model.parameters(), lr=self.init_lr, momentum=0.9, weight_decay=0.00005)

    def lr_scheduler(self, epoch):
        if epoch >= 0.5 * self.num_epochs and not self.decay_time[0]:
            self.decay_time[0] = True
            for g in self.optimizer.param_groups:
                g['lr'] = self.init_lr * self.decay_rate
        if epoch >= 0.8 * self.num_epochs and not self.decay_time[1]:
            self.decay_time[1] = True
            for g in self.optimizer.param_groups:
                g['lr'] = self.init_lr * self.decay_rate * self.decay_rate

    def _to_device(self):
        self.model.to(self.gpuid)
        if self._cl:
            self.model.to(memory_format=torch.channels_last)

    def _images(self, images):
        images = images.to(self.gpuid)
        return images.contiguous(memory_format=torch.channels_last) if self._cl else images

    def _autocast(self):
        return torch.autocast('cuda', dtype=torch.bfloat16, enabled=self._bf16)

    def run(self, global_img_feature, global_txt_feature, distill_index, global_train_loader):
        self._to_device()
        self.old_model = copy.deepcopy(self.model)
        self.old_model.eval()
        self.old_model.requires_grad_(False)                 # frozen for the round: its forwards save nothing for a backward
        self.lr_scheduler(self.cur_epoch)
        for i in range(self.local_epochs):
            self.local_epoch += 1
            self.tra(global_img_feature, global_txt_feature, distill_index, global_train_loader)
        self.test()
        if getattr(self.args, 'save_client', False):
            os.makedirs(f'./saved_clients/{self.dset_name}', exist_ok=True)
            torch.save(self.model.state_dict(),
                       f'./saved_clients/{self.dset_name}/Client{self.client_id}-model_{self.local_epoch}.pth')
        del self.old_model
        self.old_model = None
        gs = getattr(self, '_graphed_contrast', None)
        self.graph_stats = None if gs is None else {'calls': gs.calls, 'replays': gs.replays, 'failed': gs.failed}
        self._graphed_contrast = self._graphed_key = None       # the round's graph (and its private pool) goes with the old model

    # -- step 3: learning ----------------------------------------------------------------------------------------
    def _features(self, model, images, captions, caption_lens):
        if self.dset_name in IMAGE_SETS:
            with self._autocast():
                out = model(self._images(images))
            return out.float() if self._bf16 else out
        out = model(captions.to(self.gpuid), caption_lens.to(self.gpuid))
        return out.squeeze() if out.dim() > 2 else out

    def _supervised_epoch(self):
        """ClientTrainer.py:322-365: CE with the one-hot margin + centre loss on the class weights (fused, 8f-4)."""
        self.model.train()
        for i, data in enumerate(self.train_loader or []):
            self.optimizer.zero_grad()
            if self.dset_name in IMAGE_SETS:
                inputs_bt, labels_bt = data
                labels_var = labels_bt.to(self.gpuid)
                with self._autocast():
                    fvec, _, class_weight, _ = self.model(self._images(inputs_bt))
                if self._bf16:
                    fvec, class_weight = fvec.float(), class_weight.float()
            else:
                inputs_bt, labels_bt, caplens = data
                labels_var = labels_bt.to(self.gpuid)
                fvec, _, class_weight, _ = self.model(inputs_bt.to(self.gpuid).contiguous(), caplens.to(self.gpuid))
            # one-hot margin + CE + centre loss + precision@1/@k: three fused launches (csrc/supervised.hip)
            k5 = {'Cifar100': 5, 'Cifar10': 5, 'AG_NEWS': 4, 'YelpReviewPolarity': 2}[self.dset_name]
            total_loss, stats = ops.supervised_glue(fvec, labels_var, class_weight, self.inter_distance, topk=k5)
            self.top1.update(stats[3], inputs_bt.size(0))
            self.top5.update(stats[4], inputs_bt.size(0))
            self.losses.update(total_loss.detach(), inputs_bt.size(0))
            with runtime.backward_here():
                total_loss.backward()
            self.optimizer.step()
            if is_test:
                break
        # printnreset (ClientTrainer.py:308-317, called at :365): log the epoch's meters, then start fresh ones.  The meters
        # hold device scalars (no per-batch .item()); this one log line is the only host read-back of the epoch.
        if self.losses.count:
            self._log('Epoch: [{0}] {1}\tLoss {2:.4f} ({3:.4f})\tPrec@1 {4:.3f} ({5:.3f})\tPrec@5 {6:.3f} ({7:.3f})'.format(
                self.local_epoch, self.dset_name, float(self.losses.val), float(self.losses.avg), float(self.top1.val),
                float(self.top1.avg), float(self.top5.val), float(self.top5.avg)))
        self.losses, self.top1, self.top5 = AverageMeter(), AverageMeter(), AverageMeter()

    def contrast_step_fn(self, g_same, g_other, use_intra, use_inter):
        """The contrast step of this client against the round's frozen banks (ClientTrainer.py:376-421) as a function
        `step(images, captions, caption_lens, d_idx) -> detached loss`: zero_grad, features, old-model features (intra),
        inter / intra contrast, backward, SGD step -- no host synchronisation inside for an image client (d_idx an int64 device
        tensor), which is the unit the HIP graph captures (creamfl_amd/graphs.py).  `tra` iterates it over the public loader;
        bench.py --config 2 times exactly this function."""
        def step(images, captions, caption_lens, d_idx):
            self.optimizer.zero_grad(set_to_none=True)
            feature = self._features(self.model, images, captions, caption_lens)
            old_feature = None
            if use_intra:
                with torch.no_grad():
                    old_feature = self._features(self.old_model, images, captions, caption_lens)
            loss, _, _ = client_contrast_loss(feature, g_same, g_other, d_idx, old_feature,
                                              interintra_weight=self.args.interintra_weight,
                                              loss_scale=bool(self.args.loss_scale), use_inter=use_inter,
                                              use_intra=use_intra, root=True)        # the backward starts at this loss (:420)
            with runtime.backward_here():
                loss.backward()
            self.optimizer.step()
            return loss.detach()
        return step

    def tra(self, global_img_feature, global_txt_feature, distill_index, global_train_loader):
        self._supervised_epoch()
        use_intra = bool(self.args.contrast_local_intra)
        use_inter = bool(self.args.contrast_local_inter)
        if not (use_intra or use_inter):
            return
        g_img = global_img_feature.to(self.gpuid)
        g_txt = global_txt_feature.to(self.gpuid)
        is_img = self.dset_name in IMAGE_SETS
        g_same, g_other = (g_img, g_txt) if is_img else (g_txt, g_img)
        distill_dict = {b: a for a, b in enumerate(distill_index)}
        for m in ([self.model, self.old_model] if use_intra else [self.model]):
            m.phase = 'extract_conv_feature'
            m.is_train = False
        self._log('Start %s Contrasting!' % ('Intra & Inter' if use_intra and use_inter else
                                             'Intra-modal' if use_intra else 'Inter-modal'))
        self.last_contrast_loss = None

        contrast_step = self.contrast_step_fn(g_same, g_other, use_intra, use_inter)

        # The whole step replays from one HIP graph, re-captured every round (the banks, the old model and the learning rate are
        # constants of a round).  Image clients have fixed batch shapes.  Text clients: with the recurrence of gru.hip the caption
        # lengths stay on the device (no packed sequences), so the step is capturable once every batch is padded to ONE width
        # (`caption_graph_width`: padded words are masked out of the PIE head and never reached by the recurrence -- same values,
        # zero gradient); a batch wider than the captured width runs eagerly.  --client_graph 0 switches it off.
        graphed = None
        can_graph = is_img or ops.gru_last_supported(getattr(self.model, 'rnn', None))
        if can_graph and bool(int(flags.get(self.args, 'client_graph'))) and not is_test and torch.device(self.gpuid).type == 'cuda':
            # ONE capture per round: the local epochs of a round see the same banks, the same old model and the same learning
            # rate, so the later epochs replay the first one's graph (the closure it captured holds exactly those objects)
            key = (self.cur_epoch, g_same.data_ptr(), g_other.data_ptr(), id(self.old_model), use_intra, use_inter,
                   tuple(g['lr'] for g in self.optimizer.param_groups))
            graphed = getattr(self, '_graphed_contrast', None)
            if graphed is None or getattr(self, '_graphed_key', None) != key:
                fn = (lambda images, d_idx: contrast_step(images, None, None, d_idx)) if is_img else \
                    (lambda captions, caption_lens, d_idx: contrast_step(None, captions, caption_lens, d_idx))
                graphed = self._graphed_contrast = GraphedStep(fn, warmup=3, log=self._log, guard_params=[p for g in self.optimizer.param_groups for p in g['params']])
                graphed.caption_width = None
                self._graphed_key = key
        for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(global_train_loader):
            d_idx = operator.itemgetter(*index)(distill_dict)
            d_idx = d_idx if isinstance(d_idx, tuple) else (d_idx,)
            if graphed is not None and is_img:
                loss = graphed(images, torch.as_tensor(d_idx, dtype=torch.int64), device=torch.device(self.gpuid))
            elif graphed is not None:
                if graphed.caption_width is None:
                    graphed.caption_width = caption_graph_width(captions.shape[1])
                loss = graphed(pad_captions(captions, graphed.caption_width), torch.as_tensor(caption_lens, dtype=torch.int64),
                               torch.as_tensor(d_idx, dtype=torch.int64), device=torch.device(self.gpuid))
            else:
                loss = contrast_step(images, captions, caption_lens, d_idx)
            self.last_contrast_loss = loss
            if is_test:
                break
        for m in ([self.model, self.old_model] if use_intra else [self.model]):
            m.phase = 'None'
            m.is_train = True

    def test(self):
        if self.global_test_set is None:
            return
        self.model.eval()
        with torch.no_grad():
            for i, data in enumerate(self.global_test_set):
                if self.dset_name in IMAGE_SETS:
                    inputs_bt, labels_bt = data
                    with self._autocast():
                        fvec, _, _, _ = self.model(self._images(inputs_bt))
                else:
                    inputs_bt, labels_bt, caplens = data
                    fvec, _, _, _ = self.model(inputs_bt.to(self.gpuid), caplens.to(self.gpuid))
                k5 = {'Cifar100': 5, 'Cifar10': 5, 'AG_NEWS': 4, 'YelpReviewPolarity': 2}[self.dset_name]
                prec1, prec5 = accuracy(fvec.data, labels_bt, topk=(1, k5))
                self.test_top1.update(prec1[0], inputs_bt.size(0))
                self.test_top5.update(prec5[0], inputs_bt.size(0))
        self._log('TTTEST:  Epoch: [{0}] {1}\tPrec@1 {2:.3f}\tPrec@5 {3:.3f}'.format(
            self.local_epoch, self.dset_name, float(self.test_top1.avg), float(self.test_top5.avg)))
        self.test_top1, self.test_top5 = AverageMeter(), AverageMeter()
        self.model.train()

    @property
    def modalities(self):
        """Which representations generate_logits returns (host knowledge used by dist.client_plan)."""
        return ('img',) if self.dset_name in IMAGE_SETS else ('txt',)

    def generate_logits(self, dataloader, out=None):
        """ClientTrainer.py:622-629.  `out` = {'img' | 'txt': [M, D] tensor} (section 8f-3): the representations are written
        straight into it -- the rank's slice of the round's all-gather buffer, possibly bf16 -- instead of being concatenated
        into a fresh tensor; the returned dict then holds that very tensor."""
        key = 'img' if self.dset_name in IMAGE_SETS else ('txt' if self.dset_name in TEXT_SETS else None)
        assert key is not None
        vec, idx = self.extract_pub_feature(dataloader, out=None if out is None else out[key])
        return {'img': vec if key == 'img' else None, 'txt': vec if key == 'txt' else None}, idx

    def extract_pub_feature(self, dataloader, out=None):
        """ClientTrainer.py:631-664, device-resident."""
        self._to_device()
        self.model.phase = 'extract_conv_feature'
        self.model.is_train = False
        was_training = self.model.training
        feature, distill_index, off = [], [], 0
        with torch.no_grad():
            for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(dataloader):
                f = self._features(self.model, images, captions, caption_lens).detach()
                if out is None:
                    feature.append(f.float())
                else:
                    out[off:off + f.shape[0]].copy_(f)           # converts to the buffer's dtype (bf16 wire) on the way
                    off += f.shape[0]
                distill_index.extend(index)
        if out is None:
            feature = torch.cat(feature, dim=0)
        else:
            if off != out.shape[0]:
                raise RuntimeError(f'public set yielded {off} rows, the representation buffer has {out.shape[0]}')
            feature = out
        self.model.phase = 'None'
        self.model.is_train = True
        self.model.train(was_training)
        return feature, distill_index
