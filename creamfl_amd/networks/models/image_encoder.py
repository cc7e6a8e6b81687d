"""Image tower of PCME.  Mirrors src/networks/models/image_encoder.py:17-71: CNN trunk -> 7x7 map ->
avgpool + fc, PIENet over the 49 positions, (head_proj), l2-normalise.  Same attribute names
(cnn, fc, pie_net, head_proj) and the same output dict.

MI355X layout: the trunk runs channels_last, so its [N, Cd, 7, 7] output is physically
[N, 7, 7, Cd] and the reference's `.view(N, Cd, 49).transpose(1, 2)` ([N, 49, Cd], :62-64) is a free
contiguous view.  avgpool, attention pooling, sigmoid/residual/LayerNorm and the final l2-normalise
are fused into the HIP kernels of csrc/pie.hip (X is read once for both poolings).
"""
import torch.nn as nn

from ... import ops
from ..backbones import resnet_trunk
from .pie_model import PIENet


class EncoderImage(nn.Module):
    def __init__(self, config, mlp_local):
        super().__init__()
        embed_dim = config.embed_dim
        self.cnn = resnet_trunk(config.cnn_type)          # random init: no ImageNet weights offline
        cnn_dim = self.cnn_dim = self.cnn.out_dim
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))       # kept for state_dict / API parity; fused in forward
        self.fc = nn.Linear(cnn_dim, embed_dim)
        self.pie_net = PIENet(1, cnn_dim, embed_dim, cnn_dim // 2)
        for param in self.cnn.parameters():
            param.requires_grad = True
        self.n_samples_inference = config.get('n_samples_inference', 0)
        self.mlp_local = mlp_local
        if self.mlp_local:                                 # hard-wired to 512 in the reference (:42-48)
            self.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                           nn.Linear(512, 512))

    def init_weights(self):
        nn.init.xavier_uniform_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0.0)

    def head(self, fmap):
        """Everything after the trunk: avgpool + fc, PIE attention pooling over the positions, sigmoid / residual / LayerNorm,
        (head_proj), l2-normalise (image_encoder.py:55-67) on a [N, Cd, h, w] map.  Returns (embedding, attention, residual)."""
        n, cd, h, w = fmap.shape
        x = fmap.permute(0, 2, 3, 1).reshape(n, h * w, cd)                 # [N, 49, Cd]; a view under channels_last
        if not self.mlp_local:
            out, _, attn, residual = self.pie_net.forward_fused(None, x, None, l2norm=True, out_from_mean=self.fc)
        else:
            _, o, attn, residual = self.pie_net.forward_fused(None, x, None, l2norm=False, out_from_mean=self.fc)
            out = ops.l2_normalize(self.head_proj(o))
        return out, attn, residual

    def forward(self, images):
        out, _, _ = self.head(self.cnn.features(images))                   # trunk: [N, Cd, h, w]
        return {'embedding': out}
