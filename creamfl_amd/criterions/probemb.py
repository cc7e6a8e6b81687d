"""MCSoftContrastiveLoss on the HIP path (row A1).

Mirrors src/criterions/probemb.py:89-256: same constructor (config with init_shift,
init_negative_scale, num_samples, optional uniform_lambda / vib_beta), same learnable parameters
(`shift`, `negative_scale`, shape [1]), same forward signature and (loss, loss_dict) return, same
`match_prob`.  The all-pairs work (full_sampling / pairwise_sampling / batchwise_cdist /
soft_contrastive_nll, :7-86,150-208) is one fused HIP path: fp32-MFMA similarity GEMM ->
distance -> softplus NLL -> reductions, with the analytic backward as two more MFMA GEMMs
(csrc/pair_loss.hip).

CreamFL runs PCME without its probabilistic part (logsigma outputs are None, uniform_lambda =
vib_beta = 0, src/coco.yaml:41-47): those branches raise NotImplementedError here instead of
silently computing something else.
"""
import torch
import torch.nn as nn

from .. import ops


class MCSoftContrastiveLoss(nn.Module):
    def __init__(self, config, reduction='sum'):
        super().__init__()
        if reduction not in {'mean', 'sum', None}:
            raise ValueError('unknown reduction {}'.format(reduction))
        if reduction != 'sum':
            raise NotImplementedError("creamfl_amd implements reduction='sum' (the only one CreamFL uses)")
        self.reduction = reduction

        device = 'cuda:0' if torch.cuda.is_available() else 'cpu'      # probemb.py:125-126
        self.shift = nn.Parameter(config.init_shift * torch.ones(1, device=device))
        self.negative_scale = nn.Parameter(config.init_negative_scale * torch.ones(1, device=device))

        self.num_samples = config.num_samples
        self.uniform_lambda = config.get('uniform_lambda', 0)
        self.vib_beta = config.get('vib_beta', 0)
        if self.uniform_lambda != 0 or self.vib_beta != 0:
            raise NotImplementedError('uniform_lambda / vib_beta are 0 in every CreamFL config; the '
                                      'probabilistic PCME terms are outside the hot path')

    def match_prob(self, image_features, caption_features, image_logsigma, caption_logsigma,
                   use_batchwise_cdist=True):
        """probemb.py:210-219 (2-D features: one distance per row pair, broadcasting a length-1 side)."""
        diff = image_features - caption_features
        distance = torch.sqrt((diff ** 2).sum(-1) + 1e-6).reshape(max(len(image_features), len(caption_features)), -1)
        distance = distance.to(self.negative_scale.device).float()
        logits = -self.negative_scale * distance + self.shift
        prob = torch.exp(logits) / (torch.exp(logits) + torch.exp(-logits))
        return prob.mean(axis=1)

    def forward(self, image_features, caption_features, image_logsigma=None, caption_logsigma=None, **kwargs):
        if image_features.dim() != 2 or caption_features.dim() != 2:
            raise NotImplementedError('creamfl_amd handles [N, D] features (CreamFL never samples K > 1 embeddings)')
        if len(image_features) != len(caption_features):
            raise RuntimeError('# anchors ({}) != # candidates ({})'.format(image_features.shape,
                                                                           caption_features.shape))
        loss, stats = ops.pair_loss(image_features, caption_features, self.negative_scale, self.shift)
        return loss, ops.LazyLossDict(stats, self.shift, self.negative_scale)
