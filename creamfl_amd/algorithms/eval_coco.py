"""COCO retrieval evaluator on the HIP path (row A6).

Mirrors src/algorithms/eval_coco.py:74-448 (COCOEvaluator: set_model / set_criterion / set_logger /
extract_features / evaluate_recall / evaluate_n_fold / evaluate, same scores dict).  Differences by
design: features stay on the device as fp32 [n, D] (the reference keeps fp64 numpy buffers with 7
identical copies of every vector, :135-136,175,181), and evaluate_recall calls the fp64 rank-count
kernel (csrc/rank.hip) instead of mm + 7x7 fold + full sort + a python loop per query (:37-51,
:296-317); the ranks are the same integers.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def recall_at_k(ranks, k):
    """eval_coco.py:22-29"""
    return 100.0 * len(np.where(ranks < k)[0]) / len(ranks)


class COCOEvaluator(object):
    def __init__(self, eval_method='matmul', n_crossfolds=-1, extract_device='cuda', eval_device='cuda',
                 verbose=False):
        if eval_method != 'matmul':
            raise NotImplementedError("creamfl_amd evaluates with eval_method='matmul' (what CreamFL uses)")
        self.eval_method = eval_method
        self.extract_device = extract_device
        self.eval_device = eval_device
        self.logger = None
        self.n_crossfolds = n_crossfolds
        try:
            from tqdm import tqdm
            self.pbar = partial(tqdm, disable=not verbose)
        except ImportError:
            self.pbar = lambda x: x
        self.autocast_dtype = None

    def set_model(self, model):
        self.model = model
        m = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
        self.n_embeddings = m.n_embeddings
        self.feat_size = m.embed_dim

    def set_criterion(self, criterion):
        self.criterion = criterion

    def set_logger(self, logger):
        self.logger = logger

    @torch.no_grad()
    def extract_features(self, dataloader):
        """eval_coco.py:118-222 with device-resident fp32 buffers."""
        self.model.eval()
        self.model.to(self.extract_device)
        num_images = dataloader.dataset.n_images
        num_captions = len(dataloader.dataset)
        dev = torch.device(self.extract_device)
        image_features = torch.zeros(num_images, self.feat_size, device=dev)
        caption_features = torch.zeros(num_captions, self.feat_size, device=dev)
        image_classes = np.zeros(num_images)
        caption_classes = np.zeros(num_captions)
        image_ids_ = np.zeros(num_images)
        caption_ids = np.zeros(num_captions)
        cur_image_idx = 0
        cur_caption_idx = 0
        seen_image_ids = set()
        iid_to_cls = dataloader.dataset.iid_to_cls

        def get_image_class(image_id):
            return iid_to_cls.get(image_id, image_id) if iid_to_cls else image_id

        for images, captions, captions_word, caption_lens, ann_ids, image_ids, _ in self.pbar(dataloader):
            images = images.to(dev)
            captions = captions.to(dev)
            caption_lens = caption_lens.to(dev)
            with torch.autocast('cuda', dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
                output = self.model(images, captions, captions_word, caption_lens)
            _image_features = output['image_features'].float()
            _caption_features = output['caption_features'].float()
            new_rows, new_dst = [], []
            for idx, image_id in enumerate(image_ids):
                image_id = int(image_id)
                image_class = get_image_class(image_id)
                if image_id not in seen_image_ids:
                    image_ids_[cur_image_idx] = image_id
                    seen_image_ids.add(image_id)
                    image_classes[cur_image_idx] = image_class
                    new_rows.append(idx)
                    new_dst.append(cur_image_idx)
                    cur_image_idx += 1
                caption_ids[cur_caption_idx + idx] = int(ann_ids[idx])
                caption_classes[cur_caption_idx + idx] = image_class
            if new_rows:
                image_features[torch.as_tensor(new_dst, device=dev)] = _image_features[torch.as_tensor(new_rows, device=dev)]
            caption_features[cur_caption_idx:cur_caption_idx + len(image_ids)] = _caption_features
            cur_caption_idx += len(image_ids)

        if cur_image_idx != num_images:
            raise RuntimeError('unexpected error, {} != {}'.format(cur_image_idx, num_images))
        if cur_caption_idx != num_captions:
            raise RuntimeError('unexpected error, {}, {}'.format(cur_caption_idx, num_captions))
        if set(image_classes) != set(caption_classes):
            raise RuntimeError('unexpected error, I({}) != C({})'.format(set(image_classes), set(caption_classes)))
        if not iid_to_cls:
            order = np.argsort(caption_classes, kind='stable')
            # captions grouped in the order of image_classes (the reference's np.where loop, :200-207)
            sorted_caption_idx = []
            by_cls = {}
            for i in order:
                by_cls.setdefault(caption_classes[i], []).append(i)
            for image_class in image_classes:
                sorted_caption_idx.extend(by_cls[image_class])
            sorted_caption_idx = np.array(sorted_caption_idx)
            caption_ids = caption_ids[sorted_caption_idx]
            caption_classes = caption_classes[sorted_caption_idx]
            caption_features = caption_features[torch.as_tensor(sorted_caption_idx, device=dev)]
        return {
            'image_features': image_features, 'caption_features': caption_features,
            'image_sigmas': np.zeros((num_images, self.feat_size)),
            'caption_sigmas': np.zeros((num_captions, self.feat_size)),
            'image_ids': image_ids_, 'caption_ids': caption_ids,
            'image_classes': torch.from_numpy(image_classes), 'caption_classes': torch.from_numpy(caption_classes),
        }

    @torch.no_grad()
    def evaluate_recall(self, q_features, g_features, q_labels, g_labels, q_ids=None, g_ids=None, batch_size=1024):
        """eval_coco.py:273-334.  q_features [Nq, D] (or the reference's [Nq, K, D]: the K identical copies are
        collapsed), labels any array-like.  `batch_size` is accepted for signature parity; the kernel tiles
        on its own."""
        if len(q_features) != len(q_labels):
            raise RuntimeError('length mismatch {}, {}'.format(q_features.shape, q_labels.shape))
        if len(g_features) != len(g_labels):
            raise RuntimeError('length mismatch {}, {}'.format(g_features.shape, g_labels.shape))
        dev = torch.device(self.eval_device)

        def prep(f):
            f = torch.as_tensor(f)
            if f.dim() == 3:
                f = f[:, 0, :]
            return f.to(device=dev, dtype=torch.float32)

        def lab(l):
            return torch.as_tensor(np.asarray(l)).to(torch.int64)

        ranks = ops.rank_count(prep(q_features), prep(g_features), lab(q_labels), lab(g_labels))
        best_pred_ranks = ranks.cpu().numpy().astype(np.float64)
        recall_1 = recall_at_k(best_pred_ranks, 1)
        recall_5 = recall_at_k(best_pred_ranks, 5)
        recall_10 = recall_at_k(best_pred_ranks, 10)
        return {
            'recall_1': recall_1, 'recall_5': recall_5, 'recall_10': recall_10,
            'rsum': recall_1 + recall_5 + recall_10,
            'medr': np.floor(np.median(best_pred_ranks)) + 1,
            'meanr': np.mean(best_pred_ranks) + 1,
        }

    def evaluate_n_fold(self, extracted_features, n_crossfolds, n_images_per_crossfold, n_captions_per_crossfold,
                        eval_batch_size):
        """eval_coco.py:336-390"""
        image_features = extracted_features['image_features']
        caption_features = extracted_features['caption_features']
        image_classes = extracted_features['image_classes']
        caption_classes = extracted_features['caption_classes']
        keys = ['recall_1', 'recall_5', 'recall_10', 'rsum', 'medr', 'meanr']
        n_fold_scores = {'i2t': {k: [] for k in keys}, 't2i': {k: [] for k in keys}}
        for idx in range(n_crossfolds):
            if self.logger:
                self.logger.log('evaluating {}-th fold'.format(idx + 1))
            i0, i1 = idx * n_images_per_crossfold, (idx + 1) * n_images_per_crossfold
            c0, c1 = idx * n_captions_per_crossfold, (idx + 1) * n_captions_per_crossfold
            _scores = {
                'i2t': self.evaluate_recall(image_features[i0:i1], caption_features[c0:c1], image_classes[i0:i1],
                                            caption_classes[c0:c1], batch_size=eval_batch_size),
                't2i': self.evaluate_recall(caption_features[c0:c1], image_features[i0:i1], caption_classes[c0:c1],
                                            image_classes[i0:i1], batch_size=eval_batch_size),
            }
            for _task, _task_scores in _scores.items():
                for key, val in _task_scores.items():
                    n_fold_scores[_task][key].append(val)
        return {_task: {key: np.mean(np.array(val)) for key, val in _task_scores.items()}
                for _task, _task_scores in n_fold_scores.items()}

    @torch.no_grad()
    def evaluate(self, dataloader, n_crossfolds=None, n_images_per_crossfold=1000, n_captions_per_crossfold=5000,
                 eval_batch_size=1024, key=None):
        """eval_coco.py:392-448"""
        scores = {}
        if self.logger:
            self.logger.log('extracting features...')
        ef = self.extract_features(dataloader)
        scores['mean_log_image_sigma'] = np.mean(ef['image_sigmas'])
        scores['mean_log_caption_sigma'] = np.mean(ef['caption_sigmas'])
        if n_crossfolds is None:
            n_crossfolds = self.n_crossfolds
        if dataloader.dataset.iid_to_cls:
            print('"use_class" setting does not evaluate 1k crossfolds')
            n_crossfolds = -1
        if n_crossfolds > 0:
            scores['n_fold'] = self.evaluate_n_fold(ef, n_crossfolds, n_images_per_crossfold,
                                                    n_captions_per_crossfold, eval_batch_size)
        if self.logger:
            self.logger.log('evaluating i2t...')
        scores['i2t'] = self.evaluate_recall(ef['image_features'], ef['caption_features'], ef['image_classes'],
                                             ef['caption_classes'], batch_size=eval_batch_size)
        if self.logger:
            self.logger.log('evaluating t2i...')
        scores['t2i'] = self.evaluate_recall(ef['caption_features'], ef['image_features'], ef['caption_classes'],
                                             ef['image_classes'], batch_size=eval_batch_size)
        for key in ('rsum', 'medr', 'meanr'):
            scores[key] = scores['i2t'][key] + scores['t2i'][key]
        return scores
