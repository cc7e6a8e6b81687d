"""Client image encoder (row A2c): ResNet trunk (library convolutions + the fused BatchNorm / pooling kernels of
networks/backbones.py) -> global average -> x scale -> optional projection, then -- selected by the attributes ClientTrainer
mutates (`phase`, `is_train`: ClientTrainer.py:372-375) -- the l2-normalised embedding (HIP kernel), the two classifier heads
with zero-clamped weights, or the raw feature.  Behavioural contract = src/networks/resnet_client.py:102-250; parameter names
are the reference's (its checkpoints load with strict=True; tests/golden/a2c_img_*.npz come from the reference's forward)."""
import math

import torch.nn as nn

from .. import ops
from .backbones import BasicBlock, BNAct, Bottleneck, MaxPool3s2, TrunkConv, first_of, stem_tail
from .language_model import clamped_head, local_projection_head

STAGE_WIDTHS = (64, 128, 256, 512)


class ResNet(nn.Module):
    def __init__(self, block, layers, **kwargs):
        super().__init__()
        self.embed_dim = kwargs['embed_dim']
        self.is_train = bool(kwargs['is_train'])
        self.scale = int(kwargs['scale'])
        self.phase = str(kwargs.get('phase', 'none'))
        self.mlp_local = kwargs.get('mlp_local', False)
        # stem
        self.conv1 = TrunkConv(3, 64, 7, stride=2, padding=3)        # (an nn.Conv2d: same parameter name, same state_dict)
        self.bn1 = BNAct(64)                                         # (an nn.BatchNorm2d: same state_dict; ReLU fused where the fused kernels apply)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = MaxPool3s2()
        # four stages; only the first keeps the resolution
        self.inplanes = 64
        for i, (width, depth) in enumerate(zip(STAGE_WIDTHS, layers)):
            setattr(self, f'layer{i + 1}', self._make_layer(block, width, depth, stride=1 if i == 0 else 2))
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        trunk_dim = STAGE_WIDTHS[-1] * block.expansion
        if not (self.embed_dim == 512 and trunk_dim == 512):       # the reference skips the projection only for 512 -> 512
            self.linear = nn.Linear(trunk_dim, self.embed_dim)
        self.class_fc_2 = nn.Linear(self.embed_dim, kwargs['num_class'])
        self.class_fc_22 = nn.Linear(self.embed_dim, 80)
        if self.mlp_local:
            self.head_proj = local_projection_head()
        self.reset_trunk()

    def reset_trunk(self):
        """He-normal convolutions (fan-out), unit BatchNorm scale, zero BatchNorm shift (resnet_client.py:139-145)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        out_planes = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_planes:
            shortcut = nn.Sequential(TrunkConv(self.inplanes, out_planes, 1, stride=stride),
                                     BNAct(out_planes))
        stage = [block(self.inplanes, planes, stride, shortcut)] + [block(out_planes, planes) for _ in range(blocks - 1)]
        self.inplanes = out_planes
        return nn.Sequential(*stage)

    def extract_conv_feature(self, x):
        x = stem_tail(self.bn1, self.maxpool, self.conv1(x))
        for i in range(1, 5):
            x = getattr(self, f'layer{i}')(x)
        return first_of(x)

    def forward(self, x):
        feat = self.avg_pool(self.extract_conv_feature(x)).flatten(1) * self.scale
        if hasattr(self, 'linear'):
            feat = self.linear(feat)
        if self.phase == 'extract_conv_feature':
            # (with --mlp_local the reference normalises twice, resnet_client.py:186-190; the second pass is kept)
            return ops.l2_normalize(ops.l2_normalize(self.head_proj(feat)) if self.mlp_local else feat)
        if not self.is_train:
            return feat
        logits, w = clamped_head(self.class_fc_2, feat)
        logits2, w2 = clamped_head(self.class_fc_22, feat)
        return logits, logits2, w, w2


def resnet10_client(pretrained=False, **kwargs):
    return ResNet(BasicBlock, [1, 1, 1, 1], **kwargs)


def resnet18_client(pretrained=False, **kwargs):
    """`pretrained` is accepted for signature parity; ImageNet weights cannot be downloaded offline
    (load a torchvision resnet18 state_dict with strict=False to reproduce it)."""
    kwargs.pop('pool_type', None)
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet_50(pretrained=False, **kwargs):
    kwargs.pop('pool_type', None)
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
