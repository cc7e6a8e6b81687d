"""Encoder trunks in plain torch.nn (convolutions and dense GEMMs run on MIOpen / hipBLASLt; they are
below the hot path of SURVEY section 8 and are not hand-written).

torchvision is not available offline, so the ResNet definitions live here; module and parameter
names follow torchvision.models.resnet (conv1, bn1, layer{1-4}.{i}.conv{1-3}, downsample.{0,1}, fc)
so that an ImageNet state_dict loads unchanged.  `BertModel` follows the Hugging Face parameter
names (embeddings.*, encoder.layer.{i}.attention.self.{query,key,value}, ...) for the same reason,
and uses F.scaled_dot_product_attention.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class BNAct(nn.BatchNorm2d):
    """BatchNorm2d whose forward can also take the residual to add and apply the ReLU: `bn(x, residual, relu)`.
    Same parameters / buffers / state_dict as nn.BatchNorm2d.  For channels_last bf16 activations on the GPU (the
    autocast training regime of the bench) the whole normalise -> add -> relu chain and its backward run as the fused
    HBM-streaming kernels of csrc/bnorm.hip; any other input (fp32, NCHW, CPU) takes the plain library path."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # `num_batches_tracked += 1` is a one-element kernel per BN layer per step (105 launches, 0.5 ms of the
        # ResNet-101 step); with a fixed momentum nothing reads it during training, so the increments are counted
        # on the host and folded into the buffer whenever the state is observed (state_dict) or the library path runs.
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(BNAct._flush_hook)

    @staticmethod
    def _flush_hook(module, prefix, keep_vars):
        module.flush_num_batches_tracked()

    def flush_num_batches_tracked(self):
        if self._nbt_pending and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._nbt_pending)
        self._nbt_pending = 0

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, residual=None, relu=False, two=False):
        """`two=True` (block outputs): returns (y, y') -- on the fused training path one buffer under two tensor
        objects, so that the gradient from the next convolution and the one from the next residual add reach the
        fused backward separately and are summed there; otherwise the same tensor twice."""
        from .. import ops
        if self.track_running_stats and self.momentum is not None and self.affine and ops.bn_act_supported(x, self.num_features):
            if self.training:
                if torch.cuda.is_current_stream_capturing():
                    # inside a HIP-graph capture (the clients' contrast step, creamfl_amd/graphs.py) the count must live on the
                    # device: a replay runs no Python, and a host counter would stand still while the graph trains
                    # (the host's pending count stays pending: folding it in HERE would be captured and replayed too)
                    if self.num_batches_tracked is not None:
                        self.num_batches_tracked.add_(1)
                else:
                    self._nbt_pending += 1
                return ops.bn_act_train(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum,
                                        self.eps, relu=relu, residual=residual, two=two)
            if not torch.is_grad_enabled():
                y = ops.bn_act_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                    relu=relu, residual=residual)
                return (y, y) if two else y
        self.flush_num_batches_tracked()
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return (y, y) if two else y


class TrunkConv(nn.Conv2d):
    """nn.Conv2d(cin, cout, k, stride, padding, bias=False) of the ResNet trunks.  On the channels_last GPU training
    path the backward is split (ops.conv_split): data gradient on the critical path -- for 1x1 / stride-1 kernels on the
    hand-written MFMA GEMM --, weight gradient on an auxiliary stream where it overlaps the HBM-bound kernels of the
    layers below.  Forward (and everything on other inputs) is the library convolution.
    Note: outside DDP the deferred weight gradient is accumulated into `weight.grad` at the end of `backward()` (what
    AccumulateGrad does) rather than returned through autograd, so `torch.autograd.grad(..., conv.weight)` sees None;
    set CFL_NO_SIDE_WGRAD=1 for the plain behaviour."""

    bn_follows = False      # set by the block that owns the module: a BatchNorm consumes the output (ops.conv_split)

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride, padding, bias=False)

    def forward(self, x):
        if (torch.is_grad_enabled() and not _NO_CONV_SPLIT and x.is_cuda and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last)):
            from .. import ops
            if self.in_channels == 3 and ops.stem_conv_supported(x, self.weight, self.stride[0], self.padding[0]):
                return ops.stem_conv(x, self.weight, side_wgrad=not _NO_SIDE_WGRAD)      # 3-channel 7x7 stem: space-to-depth form
            if x.dtype == self.weight.dtype:
                # the weight-gradient side stream belongs to the bf16 trunks of the server step; for the clients' fp32 encoders it
                # buys nothing eager (21.6 vs 21.4 ms per contrast step) and costs their HIP graph 7 ms (28.6 vs 21.1: the
                # fork / join pattern of the flushes serialises in the replay)
                return ops.conv_split(x, self.weight, self.stride[0], self.padding[0],
                                      side_wgrad=not _NO_SIDE_WGRAD and x.dtype == torch.bfloat16,
                                      bn_follows=self.bn_follows and self.training)
        if (x.is_cuda and x.dim() == 4 and x.dtype == self.weight.dtype and self.stride[0] == self.stride[1]
                and self.padding[0] == self.padding[1] and isinstance(self.padding[0], int)):
            from .. import ops
            if ops.X3CONV[0] and not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)) \
                    and ops.conv3x3_x3_supported(x, self.weight, self.stride[0], self.padding[0]):
                # forward-only passes of an fp32 channels_last encoder (the clients' old model, representation extraction) on the
                # 3 x bf16-split kernel as well (csrc/conv3x3_x3.hip)
                return ops.conv3x3_x3_forward(x, self.weight, stride=self.stride[0])
            if ops.conv_gate_worthwhile(x, self.weight, self.stride[0], self.padding[0]):
                # every other case (the fp32 NCHW client encoders; forward-only passes: representation extraction, evaluation):
                # the library's kernels, but answered from the shipped find-db per call where it holds the problem instead of a
                # timed search per process and shape
                if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
                    return ops.conv_gated(x, self.weight, self.stride[0], self.padding[0])
                return ops._conv_fwd(x, self.weight, self.stride[0], self.padding[0])
        return super().forward(x)


class Conv1x1(TrunkConv):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1)


class MaxPool3s2(nn.MaxPool2d):
    """nn.MaxPool2d(3, 2, 1) (the ResNet stem pool); channels_last bf16 activations on the GPU take the streaming
    kernels of csrc/pool.hip, anything else the library path."""

    def __init__(self):
        super().__init__(3, 2, 1)

    def forward(self, x):
        from .. import ops
        if x.dtype == torch.bfloat16 and ops.bn_act_supported(x, x.shape[1]):
            return ops.maxpool3s2(x)
        return super().forward(x)


def stem_tail(bn, pool, x):
    """maxpool(relu(bn(x))) of a ResNet stem: BatchNorm + ReLU + max pooling in one pass per direction where the fused kernels apply
    (training mode, channels_last bf16 / fp32 on the GPU), the two modules otherwise."""
    from .. import ops
    if (bn.training and bn.track_running_stats and bn.momentum is not None and bn.affine
            and isinstance(pool, MaxPool3s2) and ops.bn_relu_maxpool_supported(x, bn.num_features)):
        if torch.cuda.is_current_stream_capturing():          # (as BNAct.forward: inside a HIP-graph capture the count lives on the device)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
        else:
            bn._nbt_pending += 1
        return ops.bn_relu_maxpool(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    return pool(bn(x, relu=True))


def first_of(x):
    """Residual blocks hand (conv input, residual input) pairs to each other; consumers outside take the first."""
    return x[0] if isinstance(x, tuple) else x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = TrunkConv(inplanes, planes, 3, stride, 1)
        self.bn1 = BNAct(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = TrunkConv(planes, planes, 3, 1, 1)
        self.bn2 = BNAct(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        x, x_res = x if isinstance(x, tuple) else (x, x)        # (conv input, residual input): see BNAct.forward(two=)
        residual = x_res if self.downsample is None else self.downsample(x_res)
        out = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(out), residual=residual, relu=True, two=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv1x1(inplanes, planes)
        self.bn1 = BNAct(planes)
        self.conv2 = TrunkConv(planes, planes, 3, stride, 1)                 # stride on the 3x3 (torchvision v1.5)
        self.bn2 = BNAct(planes)
        self.conv3 = Conv1x1(planes, planes * 4)
        self.conv3.bn_follows = True
        self.bn3 = BNAct(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        if isinstance(downsample, nn.Sequential) and len(downsample) == 2 and isinstance(downsample[0], TrunkConv) \
                and isinstance(downsample[1], BNAct):
            downsample[0].bn_follows = True
        self.stride = stride

    def forward(self, x):
        x, x_res = x if isinstance(x, tuple) else (x, x)        # (conv input, residual input): see BNAct.forward(two=)
        residual = x_res if self.downsample is None else self.downsample(x_res)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=residual, relu=True, two=True)


class ResNetTrunk(nn.Module):
    """conv1 .. layer4 of a torchvision-style ResNet; `features(x)` returns the last feature map."""

    def __init__(self, block, layers, relu_inplace=True):
        super().__init__()
        self.inplanes = 64
        self.conv1 = TrunkConv(3, 64, 7, 2, 3)
        self.bn1 = BNAct(64)
        self.relu = nn.ReLU(inplace=relu_inplace)
        self.maxpool = MaxPool3s2()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.out_dim = 512 * block.expansion
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                TrunkConv(self.inplanes, planes * block.expansion, 1, stride),
                BNAct(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def features(self, x):
        x = stem_tail(self.bn1, self.maxpool, self.conv1(x))
        return first_of(self.layer4(self.layer3(self.layer2(self.layer1(x)))))

    def forward(self, x):
        return self.features(x)


_RESNETS = {
    'resnet10': (BasicBlock, [1, 1, 1, 1]), 'resnet18': (BasicBlock, [2, 2, 2, 2]),
    'resnet34': (BasicBlock, [3, 4, 6, 3]), 'resnet50': (Bottleneck, [3, 4, 6, 3]),
    'resnet101': (Bottleneck, [3, 4, 23, 3]), 'resnet152': (Bottleneck, [3, 8, 36, 3]),
}


_VIT_NO_FUSE = bool(os.environ.get('CFL_NO_VIT_FUSE'))       # A/B: the ViT blocks on aten LayerNorm / GELU / adds


class _ViTBlock(nn.Module):
    def __init__(self, dim, heads, mlp_dim):
        super().__init__()
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.heads = heads
        self.in_proj = nn.Linear(dim, 3 * dim)
        self.out_proj = nn.Linear(dim, dim)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Linear(mlp_dim, dim))

    def forward(self, x):
        B, L, D = x.shape
        q, k, v = self.in_proj(self.ln_1(x)).view(B, L, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, D)
        x = x + self.out_proj(a)
        return x + self.mlp(self.ln_2(x))

    def forward_fused(self, x, h, next_ln):
        """The same block on the glue kernels of the BERT tower (csrc/bertfuse.hip; round 6).  x: the residual stream (bf16), h =
        ln_1(x), already formed by the previous block's tail; returns (x', next_ln(x')).  Each residual add + bias + the LayerNorm that
        reads the sum is ONE pass per direction (`ops.preln_add_layernorm`: the backward also sums the two gradients of the residual
        stream and yields the bias / LayerNorm parameter gradients), bias + GELU another (`ops.bert_bias_gelu`); the GEMMs stay on
        hipBLASLt, the attention on the library's SDPA.  The residual stream is held in bf16 (eager autocast: fp32)."""
        from .. import ops
        B, L, D = x.shape
        q, k, v = self.in_proj(h).view(B, L, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, D)
        x1, h2 = ops.preln_add_layernorm(F.linear(a, self.out_proj.weight), self.out_proj.bias, x, self.ln_2.weight, self.ln_2.bias,
                                         self.ln_2.eps)
        u = ops.bert_bias_gelu(F.linear(h2, self.mlp[0].weight), self.mlp[0].bias)
        return ops.preln_add_layernorm(F.linear(u, self.mlp[2].weight), self.mlp[2].bias, x1, next_ln.weight, next_ln.bias, next_ln.eps)


class ViTTrunk(nn.Module):
    """ViT-B/16-style image trunk (BASELINE.json configs[4], a build-defined encoder: the reference only ships
    ResNets).  `features(x)` returns the patch tokens as a [N, D, H/16, W/16] map whose memory is [N, P, D], so the
    PIE head sees the same [N, P, Cd] layout as with a channels_last ResNet."""

    def __init__(self, dim=768, depth=12, heads=12, mlp_dim=3072, patch=16, img=224):
        super().__init__()
        self.patch = patch
        self.conv_proj = nn.Conv2d(3, dim, patch, patch)
        self.pos_embedding = nn.Parameter(torch.zeros(1, (img // patch) ** 2, dim).normal_(std=0.02))
        self.layers = nn.ModuleList([_ViTBlock(dim, heads, mlp_dim) for _ in range(depth)])
        self.ln = nn.LayerNorm(dim, eps=1e-6)
        self.out_dim = dim

    def patch_embed(self, x):
        """The patch projection -- a convolution whose stride equals its kernel -- as ONE GEMM over non-overlapping patches:
        [N * P, p * p * 3] x [p * p * 3, D].  As a convolution MIOpen has no tuned kernel for 16 x 16 / stride 16 on 3
        channels in NHWC bf16 and falls back to `naive_conv_*`: 22.6 ms forward + 12.2 ms weight gradient per step at
        batch 64 -- half of the configs[4] server step (rocprofv3, profiles/r3_config4_*).  Patches are taken in (kh, kw,
        c) order, the memory order of a channels_last image and of the channels_last weight, so both reshapes are cheap;
        `conv_proj` keeps its torchvision parameter names and shapes."""
        n, c, H, W = x.shape
        p = self.patch
        if H % p or W % p:
            return self.conv_proj(x).flatten(2).transpose(1, 2)
        h, w = H // p, W // p
        xp = x.permute(0, 2, 3, 1).reshape(n, h, p, w, p, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h * w, p * p * c)
        wm = self.conv_proj.weight.permute(0, 2, 3, 1).reshape(self.conv_proj.out_channels, p * p * c)
        return F.linear(xp, wm, self.conv_proj.bias), h, w

    def features(self, x):
        t = self.patch_embed(x)                                  # [N, P, D]
        if isinstance(t, tuple):
            t, h, w = t
        else:
            h, w = x.shape[2] // self.patch, x.shape[3] // self.patch
        n, _, d = t.shape
        pos = self.pos_embedding
        if pos.shape[1] != h * w:                                # other resolutions: bilinear resize of the grid
            g = int(pos.shape[1] ** 0.5)
            pos = F.interpolate(pos.reshape(1, g, g, d).permute(0, 3, 1, 2), size=(h, w), mode='bilinear',
                                align_corners=False).permute(0, 2, 3, 1).reshape(1, h * w, d)
        t = t + pos
        blk0 = self.layers[0]
        i8 = blk0.mlp[0].weight.shape[0] >> 3                   # (cfl_bias_gelu_bwd's column plan: >= 256 16-byte groups, or a divisor of 256)
        if (not _VIT_NO_FUSE and _bert_fusable(t, blk0.out_proj.weight, max_out=2048) and _bert_fusable(t, blk0.mlp[0].weight)
                and (i8 >= 256 or 256 % i8 == 0)
                and isinstance(blk0.mlp[1], nn.GELU) and getattr(blk0.mlp[1], 'approximate', 'none') == 'none'):
            # pre-LN chain on the fused glue: block i's tail forms block i + 1's ln_1 (the last one: the trunk's final LayerNorm)
            t = t.to(torch.bfloat16)
            hN = blk0.ln_1(t).to(torch.bfloat16)
            for i, blk in enumerate(self.layers):
                nxt = self.layers[i + 1].ln_1 if i + 1 < len(self.layers) else self.ln
                t, hN = blk.forward_fused(t, hN, nxt)
            t = hN
        else:
            for blk in self.layers:
                t = blk(t)
            t = self.ln(t)
        return t.transpose(1, 2).reshape(n, d, h, w)             # a view: memory stays [N, P, D]

    def forward(self, x):
        return self.features(x)


_VITS = {'vit_b_16': dict(dim=768, depth=12, heads=12, mlp_dim=3072), 'vit_tiny_16': dict(dim=192, depth=2, heads=3, mlp_dim=384)}


def resnet_trunk(name):
    """Image trunk by name: torchvision-style ResNets (what the reference uses) or the build-defined ViTs."""
    if name in _VITS:
        return ViTTrunk(**_VITS[name])
    if name not in _RESNETS:
        raise ValueError(f'unknown cnn_type {name}')
    block, layers = _RESNETS[name]
    return ResNetTrunk(block, layers)


# ----------------------------------------------------------------------------------------- BERT
class BertConfig:
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1):
        self.__dict__.update(locals())
        del self.__dict__['self']


BERT_CONFIGS = {
    'bert-base-uncased': dict(),
    'bert-large-uncased': dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
    'bert-mini': dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024),
}


_NO_ATTN_SMALL = bool(os.environ.get('CFL_NO_ATTN_SMALL'))       # A/B switches for measurements
_NO_CONV_SPLIT = bool(os.environ.get('CFL_NO_CONV_SPLIT'))
_NO_SIDE_WGRAD = bool(os.environ.get('CFL_NO_SIDE_WGRAD'))


def _bert_fusable(x, weight, max_out=1 << 30):
    """The fused glue kernels (csrc/bertfuse.hip) take bf16 GEMM outputs on the GPU: bf16 weights, or bf16 autocast."""
    if not (x.is_cuda and weight.shape[0] % 8 == 0 and weight.shape[0] <= max_out):
        return False
    if weight.dtype == torch.bfloat16 and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled('cuda')):
        return True
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


class PackPlan:
    """Index tensors of a packed text-tower run (BertModel.pack_plan): tok_idx [T] flat positions b L + t of the real tokens in the
    padded [B, L] frame, pos_ids [T] their positions t, cu [B + 1] int32 row offsets of the sequences, cls_rows [B] = cu[:-1]."""
    __slots__ = ('B', 'L', 'T', 'tok_idx', 'pos_ids', 'cu', 'cls_rows')


def _bert_fusable_model(m):
    """every layer of this BertModel takes the fused bf16 kernels (bf16 weights, or bf16 autocast): the packed run's condition"""
    w = m.encoder.layer[0].attention.self.query.weight
    if not w.is_cuda or w.shape[0] % 8 or (w.shape[0] // m.config.num_attention_heads) != 64 or _NO_ATTN_SMALL:
        return False
    return w.dtype == torch.bfloat16 or (torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16)


class _BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, pack=None):
        if pack is not None:            # packed tokens [T]: the real tokens of the batch only, each with its own position
            x = self.word_embeddings(input_ids.reshape(-1).index_select(0, pack.tok_idx)) + self.position_embeddings(pack.pos_ids)
            return self.dropout(self.LayerNorm(x + self.token_type_embeddings.weight[0]))
        L = input_ids.shape[1]
        pos = torch.arange(L, device=input_ids.device)
        x = self.word_embeddings(input_ids) + self.position_embeddings(pos)[None]
        tt = self.token_type_embeddings.weight[0] if token_type_ids is None else self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(x + tt))


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.h = c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x, mask, cls_only=False, pack=None):
        if pack is not None:
            return self._forward_packed(x, mask, cls_only, pack)
        B, L, H = x.shape
        def split(t):
            return t.view(B, t.shape[1], self.h, H // self.h).transpose(1, 2)
        xq = x[:, :1] if cls_only else x                   # only the [CLS] query is consumed downstream
        if not cls_only and _bert_fusable(x, self.query.weight):
            # one [3H, H] GEMM instead of three (and one dX GEMM, one bias-gradient reduction, no dX adds in backward)
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0)
            b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0)
            qkv = F.linear(x, w, b)
            from .. import ops
            if qkv.dtype == torch.bfloat16 and ops.bert_attention_supported(L, H // self.h) and not _NO_ATTN_SMALL:
                # short captions: one wavefront per (batch, head), MFMA, no saved probabilities (csrc/attn_small.hip)
                km = None if mask is None else getattr(mask, '_cfl_u8', None)
                if mask is not None and km is None:             # converted once per forward, reused by every layer
                    km = mask._cfl_u8 = mask.reshape(B, L).to(torch.uint8).contiguous()
                return ops.bert_attention(qkv, km, self.h)
            q, k, v = qkv.split(H, dim=-1)
        else:
            q, k, v = self.query(xq), self.key(x), self.value(x)
        o = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=mask)
        return o.transpose(1, 2).reshape(B, xq.shape[1], H)


    def _forward_packed(self, x, mask, cls_only, pack):
        """x: [T, H], the batch's real tokens (BertModel.pack_plan).  Full layers: the fused [3H, H] projection + the varlen form of
        csrc/attn_small.hip.  The last layer of a `cls_only` call: the [CLS] rows query keys / values scattered back into the padded
        [B, L, H] frame -- B queries, not worth a kernel of its own."""
        from .. import ops
        T, H = x.shape
        if not cls_only:
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0)
            b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0)
            return ops.bert_attention_varlen(F.linear(x, w, b), pack.cu, self.h)
        B, L = pack.B, pack.L
        q = self.query(x.index_select(0, pack.cls_rows))
        k, v = self.key(x), self.value(x)

        def padded(t):
            return t.new_zeros(B * L, H).index_copy(0, pack.tok_idx, t).view(B, L, self.h, H // self.h).transpose(1, 2)
        o = F.scaled_dot_product_attention(q.view(B, 1, self.h, H // self.h).transpose(1, 2), padded(k), padded(v), attn_mask=mask)
        return o.transpose(1, 2).reshape(B, H)


class _SelfOutput(nn.Module):
    def __init__(self, c, d_in):
        super().__init__()
        self.dense = nn.Linear(d_in, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)

    def forward(self, h, residual):
        """Returns (input of the next GEMM, next residual): one buffer under two tensor objects on the fused path
        (csrc/bertfuse.hip sums their gradients in its backward), the same tensor twice on the library path."""
        if _bert_fusable(h, self.dense.weight, max_out=2048):
            from .. import ops
            g = F.linear(h, self.dense.weight)                        # bias joins the fused kernel
            return ops.bert_dropout_add_layernorm(g, self.dense.bias, residual, self.LayerNorm.weight, self.LayerNorm.bias,
                                                  self.dropout.p if self.training else 0.0, self.LayerNorm.eps)
        y = self.LayerNorm(self.dropout(self.dense(h)) + residual)
        return y, y


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c, c.hidden_size)

    def forward(self, x, res, mask, cls_only=False, pack=None):
        if pack is not None:
            return self.output(self.self(x, mask, cls_only, pack), res.index_select(0, pack.cls_rows) if cls_only else res)
        return self.output(self.self(x, mask, cls_only), res[:, :1] if cls_only else res)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)

    def forward(self, x):
        if _bert_fusable(x, self.dense.weight):
            from .. import ops
            return ops.bert_bias_gelu(F.linear(x, self.dense.weight), self.dense.bias)
        return F.gelu(self.dense(x))


class _BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _SelfOutput(c, c.intermediate_size)

    def forward(self, x, res, mask, cls_only=False, pack=None):
        x, res = self.attention(x, res, mask, cls_only, pack)
        return self.output(self.intermediate(x), res)


class _BertEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(c) for _ in range(c.num_hidden_layers)])

    def forward(self, x, mask, cls_only=False, pack=None):
        last = len(self.layer) - 1
        res = x
        for i, l in enumerate(self.layer):
            x, res = l(x, res, mask, cls_only and i == last, pack)
        return x


class BertModel(nn.Module):
    """Minimal BERT encoder returning {'last_hidden_state': [B, L, H]} (all PCME reads is [:, 0, :],
    src/networks/models/pcme.py:44).  Randomly initialised (no checkpoint download offline)."""

    def __init__(self, name_or_config='bert-base-uncased'):
        super().__init__()
        c = name_or_config if isinstance(name_or_config, BertConfig) else BertConfig(**BERT_CONFIGS[name_or_config])
        self.config = c
        self.embeddings = _BertEmbeddings(c)
        self.encoder = _BertEncoder(c)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(0.0, 0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.LayerNorm):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, name):
        return cls(name)

    @staticmethod
    def pack_plan(host_lengths, L, device):
        """The index tensors of a PACKED run from the captions' lengths as HOST integers (no device synchronisation: the loaders
        know the lengths on the host, src/datasets/_dataloader.py:49-64 builds `cap_lengths` from python lists).  None when a
        sequence does not fit the varlen attention kernel (> 32 tokens) or the lengths do not fit the padded frame."""
        lens = [int(v) for v in host_lengths]
        if not lens or min(lens) < 1 or max(lens) > min(L, 32):
            return None
        B = len(lens)
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + n)
        tok = torch.tensor([b * L + t for b, n in enumerate(lens) for t in range(n)], dtype=torch.int64)
        plan = PackPlan()
        plan.B, plan.L, plan.T = B, L, cu[-1]
        plan.tok_idx = tok.to(device)
        plan.pos_ids = (tok % L).to(device)
        plan.cu = torch.tensor(cu, dtype=torch.int32).to(device)
        plan.cls_rows = torch.tensor(cu[:-1], dtype=torch.int64).to(device)
        return plan

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, cls_only=False, pack=None, **_):
        """`cls_only=True`: the last layer is evaluated for the [CLS] position only and `last_hidden_state` is
        [B, 1, H] -- identical values and gradients for everything PCME consumes (it reads [:, 0, :] only,
        src/networks/models/pcme.py:44), 1/12 less work in the tower.
        `pack` (a `pack_plan` of this batch; round 6): the tower runs on the batch's REAL tokens only, [T, H] -- the padded positions
        of the reference's [B, L] frame (a third of a COCO-shaped batch) are masked out of every attention and never read by the
        head, so nothing that is consumed changes (tests/test_gpu_bert.py) and a third of the tower's work goes away.  Taken on
        the fused bf16 path only; without it, or with token types, the padded frame runs."""
        mask = None
        if attention_mask is not None:
            mask = attention_mask[:, None, None, :].to(torch.bool)
        if pack is not None and (token_type_ids is not None or attention_mask is None or not input_ids.is_cuda
                                 or not _bert_fusable_model(self) or tuple(input_ids.shape) != (pack.B, pack.L)):
            pack = None
        x = self.embeddings(input_ids, token_type_ids, pack)
        if pack is not None and not (x.dtype == torch.bfloat16 or torch.is_autocast_enabled('cuda')):
            raise RuntimeError('packed BERT run outside the bf16 path')
        h = self.encoder(x, mask, cls_only, pack)
        if pack is not None:
            h = h[:, None, :] if cls_only else h.new_zeros(pack.B * pack.L, h.shape[-1]).index_copy(0, pack.tok_idx, h).view(
                pack.B, pack.L, -1)
        return {'last_hidden_state': h}
