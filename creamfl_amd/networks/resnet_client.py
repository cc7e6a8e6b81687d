"""Client image encoder (row A2c).  Mirrors src/networks/resnet_client.py:102-250: the client ResNet with
`scale`, the `phase == 'extract_conv_feature'` embedding path (l2-normalised, HIP kernel) and the classifier
heads with ReLU-clamped weights.  Mode is switched by mutating `model.phase` / `model.is_train`, exactly as
ClientTrainer does (ClientTrainer.py:372-375)."""
import math

import torch.nn as nn

from .. import ops
from .backbones import BasicBlock, BNAct, Bottleneck, MaxPool3s2, first_of


class ResNet(nn.Module):
    def __init__(self, block, layers, **kwargs):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = MaxPool3s2()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.embed_dim = kwargs['embed_dim']
        feat = 512 * block.expansion
        if kwargs['embed_dim'] != 512 or feat != 512:
            self.linear = nn.Linear(feat, self.embed_dim)
        self.class_fc_2 = nn.Linear(self.embed_dim, kwargs['num_class'])
        self.class_fc_22 = nn.Linear(self.embed_dim, 80)
        self.is_train = bool(kwargs['is_train'])
        self.scale = int(kwargs['scale'])
        self.phase = str(kwargs['phase']) if 'phase' in kwargs.keys() else 'none'
        self.mlp_local = kwargs['mlp_local'] if 'mlp_local' in kwargs.keys() else False
        if self.mlp_local:
            self.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                           nn.Linear(512, 512))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BNAct(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def extract_conv_feature(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return first_of(self.layer4(self.layer3(self.layer2(self.layer1(x)))))

    def forward(self, x):
        x = self.extract_conv_feature(x)
        avg_x = self.avg_pool(x)
        x = avg_x.view(avg_x.size(0), -1)
        x = x * self.scale
        if hasattr(self, 'linear'):
            x = self.linear(x)
        if self.phase == 'extract_conv_feature':
            if self.mlp_local:
                x = self.head_proj(x)
                x = ops.l2_normalize(x)
            return ops.l2_normalize(x)
        if self.is_train:
            fc_weight_relu = self.relu(self.class_fc_2.weight)
            self.class_fc_2.weight.data = fc_weight_relu
            fc_weight_relu2 = self.relu(self.class_fc_22.weight)
            self.class_fc_22.weight.data = fc_weight_relu2
            return self.class_fc_2(x), self.class_fc_22(x), fc_weight_relu, fc_weight_relu2
        return x


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
