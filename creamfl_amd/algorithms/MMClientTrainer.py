"""Multi-modal client.  Mirrors src/algorithms/MMClientTrainer.py:89-370: `run(...)`, `train_epoch(...)`,
`generate_logits(dataloader)` -> ({'img': [M,D], 'txt': [M,D]}, index)."""
import copy
import operator
import os

import torch
import torch.nn as nn

from .. import runtime
from ..graphs import GraphedStep
from ..utils.prefetch import DevicePrefetcher
from .base import EngineBase
from .ClientTrainer import caption_graph_width, pad_captions
from .contrast import mm_client_contrast_loss
from .optimizers import AdamP

is_test = False
# the multi-modal client's image tower in channels_last at fp32 (with the fused fp32 BatchNorm kernels and no weight-gradient side
# stream): 37.6 -> 30.4 ms per contrast step of the PCME-small model; CFL_MM_CHANNELS_LAST=0 restores NCHW
MM_CHANNELS_LAST = [os.environ.get('CFL_MM_CHANNELS_LAST', '1') == '1']


class MMClientTrainer(EngineBase):

    def run(self, global_img_feature, global_txt_feature, distill_index, global_train_loader, prefix=''):
        self._to_device()
        self.old_model = copy.deepcopy(self.model)
        self.old_model.eval()
        self.old_model.requires_grad_(False)                 # frozen for the round: its forwards save nothing for a backward
        if self.local_epoch == 0 and self.config.train.get('use_fp16'):
            self.to_half()
        self.model.train()
        for i in range(self.local_epochs):
            self.local_epoch += 1
            if self.logger is not None:
                self.logger.log(f"Epoch {self.local_epoch}")
            self.train_epoch(global_img_feature, global_txt_feature, distill_index, global_train_loader, prefix='')
        if getattr(self.args, 'save_client', False):
            os.makedirs('./saved_clients/Flicker30K', exist_ok=True)
            torch.save(self.model.state_dict(),
                       f'./saved_clients/Flicker30K/Client{self.client}-model_{self.local_epoch}.pth')
        gs = getattr(self, '_graphed_contrast', None)
        self.graph_stats = None if gs is None else {'calls': gs.calls, 'replays': gs.replays, 'failed': gs.failed}
        self._graphed_contrast = self._graphed_key = None       # the round's graph (and its private pool) goes with the old model
        del self.old_model
        self.old_model = None

    def _to_device(self):
        """Model and criterion on the device, the image tower in the layout / precision the build flags ask for."""
        from .. import flags
        self.model.to(self.device)
        self.criterion.to(self.device)
        if torch.device(self.device).type == 'cuda':
            if int(flags.get(self.args, 'client_bf16')) and self.autocast_dtype is None:
                self.to_half()                                    # opt-in, below the reference's fp32 clients (flags.py)
            elif int(flags.get(self.args, 'client_channels_last')) and MM_CHANNELS_LAST[0]:
                self.model.to(memory_format=torch.channels_last)
                self._cl = True
            if int(flags.get(self.args, 'client_conv_x3')):
                from .. import ops
                ops.X3CONV[0] = True

    def _forward(self, model, images, captions, captions_word, caption_lens):
        if (getattr(self, '_cl', False) or self.autocast_dtype is not None) and images.is_cuda:
            images = images.contiguous(memory_format=torch.channels_last)
        with torch.autocast('cuda', dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            return model(images, captions, captions_word, caption_lens)

    def _step(self, loss):
        self.optimizer.zero_grad()
        with runtime.backward_here():
            loss.backward()
        clip = self.config.train.grad_clip
        if isinstance(self.optimizer, AdamP):
            self.optimizer.step(clip=(self.model.parameters(), clip) if clip > 0 else None)
        else:
            if clip > 0:
                nn.utils.clip_grad.clip_grad_norm_(self.model.parameters(), clip)
            self.optimizer.step()

    def _local_epoch(self):
        """Local PCME training on the client's own pairs (MMClientTrainer.py:118-143).  A method of its own on purpose: its last
        `loss` / `output` must be dead before the contrast step is captured -- a live loss keeps its autograd graph's AccumulateGrad
        nodes alive, those are bound to the stream they were made on (here the default stream), the captured backward would make
        THAT stream wait for the capturing one, and the HIP runtime faults in hipStreamEndCapture when the legacy default stream
        is pulled into a capture (docs/history/tools/mm_graph_probe.py)."""
        for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(self.train_loader or []):
            images, captions, caption_lens = images.to(self.device), captions.to(self.device), caption_lens.to(self.device)
            output = self._forward(self.model, images, captions, captions_word, caption_lens)
            loss, loss_dict = self.criterion(**output)
            self._step(loss)
            if is_test:
                break

    def train_epoch(self, global_img_feature, global_txt_feature, distill_index, global_train_loader, prefix=''):
        self._local_epoch()
        use_intra = bool(self.args.contrast_local_intra)
        use_inter = bool(self.args.contrast_local_inter)
        if not (use_intra or use_inter):
            return
        g_img, g_txt = global_img_feature.to(self.device), global_txt_feature.to(self.device)
        distill_dict = {b: a for a, b in enumerate(distill_index)}
        self.last_contrast_loss = None
        contrast_step = self.contrast_step_fn(g_img, g_txt, use_intra, use_inter)
        graphed = self._graphed_for_round(contrast_step, g_img, g_txt, use_intra, use_inter)
        for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(DevicePrefetcher(global_train_loader, self.device)):
            d_idx = operator.itemgetter(*index)(distill_dict)
            d_idx = d_idx if isinstance(d_idx, tuple) else (d_idx,)
            if graphed is not None:
                if graphed.caption_width is None:
                    graphed.caption_width = caption_graph_width(captions.shape[1])
                self.last_contrast_loss = graphed(images, pad_captions(captions, graphed.caption_width),
                                                  torch.as_tensor(caption_lens, dtype=torch.int64),
                                                  torch.as_tensor(d_idx, dtype=torch.int64), device=torch.device(self.device))
            else:
                self.last_contrast_loss = contrast_step(images, captions, captions_word, caption_lens, d_idx)
            if is_test:
                break

    def _graphed_for_round(self, contrast_step, g_img, g_txt, use_intra, use_inter):
        """The contrast step as ONE HIP graph per round (banks, old model and learning rate are constants of a round; the local
        epochs of a round replay the first one's graph), like the uni-modal clients (ClientTrainer.tra): GRU text tower with the
        caption lengths on the device (gru.hip), captions padded to one width, and the fused AdamP with its step count on the
        device (AdamP.prepare_capture).  None when the step is not capturable (BERT text tower with a tokenizer, another
        optimizer, the CPU) or switched off (--client_graph 0 / --mm_client_graph 0)."""
        from .. import flags, ops
        if is_test or torch.device(self.device).type != 'cuda' or not isinstance(self.optimizer, AdamP):
            return None
        if not (int(flags.get(self.args, 'client_graph')) and int(flags.get(self.args, 'mm_client_graph'))):
            return None
        if not (self.model.config.not_bert and ops.gru_last_supported(getattr(self.model.txt_enc, 'rnn', None))):
            return None
        key = (id(self.old_model), g_img.data_ptr(), g_txt.data_ptr(), use_intra, use_inter,
               tuple(g['lr'] for g in self.optimizer.param_groups))
        graphed = getattr(self, '_graphed_contrast', None)
        if graphed is None or getattr(self, '_graphed_key', None) != key:
            self._graphed_contrast = None                                    # (the old graph first: its pinned tables return)
            log = (lambda m: self.logger.log(m)) if self.logger is not None else None
            graphed = self._graphed_contrast = GraphedStep(
                lambda images, captions, lens, d_idx: contrast_step(images, captions, None, lens, d_idx),
                warmup=3, log=log, optimizer=self.optimizer, other_threads=True)
            graphed.caption_width = None
            self._graphed_key = key
        return graphed

    def contrast_step_fn(self, g_img, g_txt, use_intra, use_inter):
        """The multi-modal client's contrast step against the round's frozen banks (MMClientTrainer.py:150-224) as a function
        `step(images, captions, captions_word, caption_lens, d_idx) -> detached loss`: both towers forward, the old model's
        forward (intra), the stacked intra / summed inter terms, backward, clip + optimizer.  `train_epoch` iterates it over the
        public loader; bench.py --config 2 times exactly this function."""
        def step(images, captions, captions_word, caption_lens, d_idx):
            images, captions, caption_lens = images.to(self.device), captions.to(self.device), caption_lens.to(self.device)
            output = self._forward(self.model, images, captions, captions_word, caption_lens)
            out_img, out_txt = output['image_features'], output['caption_features']
            old_img = old_txt = None
            if use_intra:
                with torch.no_grad():
                    output_o = self._forward(self.old_model, images, captions, captions_word, caption_lens)
                    old_img, old_txt = output_o['image_features'], output_o['caption_features']
            loss, _, _ = mm_client_contrast_loss(out_img, out_txt, g_img, g_txt, d_idx, old_img, old_txt,
                                                 interintra_weight=self.args.interintra_weight,
                                                 loss_scale=bool(self.args.loss_scale), use_inter=use_inter,
                                                 use_intra=use_intra, root=True)     # _step backpropagates from this loss
            self._step(loss)
            return loss.detach()
        return step

    modalities = ('img', 'txt')          # generate_logits returns both representations (dist.client_plan)

    def generate_logits(self, dataloader, out=None):
        """MMClientTrainer.py:326-359.  `out` = {'img': [M, D], 'txt': [M, D]}: write the representations straight into these
        (the rank's slices of the round's all-gather buffer, section 8f-3) instead of concatenating fresh tensors."""
        self._to_device()
        was_training = self.model.training
        self.model.eval()
        img_vec, txt_vec, distill_index, off = [], [], [], 0
        D = self.args.feature_dim
        with torch.no_grad():
            for idx, (images, captions, captions_word, caption_lens, _, _, index) in enumerate(dataloader):
                output = self._forward(self.model, images.to(self.device), captions.to(self.device), captions_word,
                                       caption_lens.to(self.device))
                fi, ft = output['image_features'].view(-1, D), output['caption_features'].view(-1, D)
                if out is None:
                    img_vec.append(fi.float())
                    txt_vec.append(ft.float())
                else:
                    out['img'][off:off + fi.shape[0]].copy_(fi)
                    out['txt'][off:off + ft.shape[0]].copy_(ft)
                    off += fi.shape[0]
                distill_index.extend(index)
                if is_test and idx == 1:
                    break
        self.model.train(was_training)
        if out is not None:
            if off != out['img'].shape[0] or off != out['txt'].shape[0]:
                # a short public loader (or the is_test early break) would leave stale rows of a previous round / client in the
                # all-gather buffer, and con_w would aggregate them as if they were this client's (ClientTrainer raises too)
                raise RuntimeError(f"public set yielded {off} rows, the representation buffers have "
                                   f"{out['img'].shape[0]} / {out['txt'].shape[0]}")
            return {'img': out['img'], 'txt': out['txt']}, distill_index
        return {'img': torch.cat(img_vec, dim=0).view(-1, D), 'txt': torch.cat(txt_vec, dim=0).view(-1, D)}, distill_index
