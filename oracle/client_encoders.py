"""Oracle for row A2c (client encoders) and the tower glue of row A2.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference:
  src/networks/resnet_client.py:33-66     BasicBlock
  src/networks/resnet_client.py:162-201   ResNet.extract_conv_feature / forward (both phases)
  src/networks/language_model.py:93-130   EncoderText.forward (is_train heads | l2norm path)
  src/networks/models/caption_encoder.py:87-116   GRU text tower (l2norm BEFORE head_proj)
  src/networks/models/image_encoder.py:54-71      image tower after the trunk
  src/networks/models/pcme.py:35-57       PCME.forward (10-key dict)
All functions are purely functional over a state dict with the reference's key names; library pieces below the hot
path (convolution, BatchNorm, GRU cell) are torch's CPU kernels.
Pinned by tests/golden/a2c_*.npz and tower_*.npz, which tests/golden/make_golden.py produced by running the
reference's own modules.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .pie import pie_head, l2_normalize


def _bn(sd, p, x, train, stats_out):
    rm, rv = sd[p + '.running_mean'].clone(), sd[p + '.running_var'].clone()
    y = F.batch_norm(x, rm, rv, sd[p + '.weight'], sd[p + '.bias'], training=train, momentum=0.1, eps=1e-5)
    stats_out[p + '.running_mean'], stats_out[p + '.running_var'] = rm, rv
    return y


def _basic_block(sd, p, x, stride, train, stats_out):
    """resnet_client.py:33-66: conv3x3-bn-relu-conv3x3-bn (+ downsample(x)) -relu."""
    out = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1), train, stats_out))
    out = _bn(sd, p + '.bn2', F.conv2d(out, sd[p + '.conv2.weight'], None, 1, 1), train, stats_out)
    if (p + '.downsample.0.weight') in sd:
        x = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0), train, stats_out)
    return F.relu(out + x)


def resnet_client_forward(sd, x, phase, is_train=True, train_mode=True, scale=128, layers=(1, 1, 1, 1), mlp_local=False):
    """resnet_client.py:162-201 with BasicBlock layers.  sd: {key: tensor} (tensors may require grad).
    Returns (outputs, new_sd_entries): outputs = feat [B, D] for phase == 'extract_conv_feature', else the
    (x1, x2, relu(W), relu(W2)) tuple of the supervised phase; new_sd_entries = the tensors the forward MUTATES
    (BatchNorm running statistics in train mode, the ReLU-clamped classifier weights :192-196)."""
    new = {}
    x = F.conv2d(x, sd['conv1.weight'], None, 2, 3)
    x = F.relu(_bn(sd, 'bn1', x, train_mode, new))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblk in enumerate(layers, start=1):
        for b in range(nblk):
            x = _basic_block(sd, f'layer{li}.{b}', x, 2 if (li > 1 and b == 0) else 1, train_mode, new)
    x = x.mean(dim=(2, 3))                              # AdaptiveAvgPool2d((1, 1)) + view :177-178
    x = x * scale                                       # :179
    if 'linear.weight' in sd:                           # embed_dim != 512 :181-182
        x = F.linear(x, sd['linear.weight'], sd['linear.bias'])
    if phase == 'extract_conv_feature':                 # :184-189
        if mlp_local:
            raise NotImplementedError('head_proj fixture not generated')
        return F.normalize(x, p=2, dim=1), new
    if is_train:                                        # :192-200
        w = F.relu(sd['class_fc_2.weight'])
        w2 = F.relu(sd['class_fc_22.weight'])
        new['class_fc_2.weight'], new['class_fc_22.weight'] = w.detach(), w2.detach()
        # after `weight.data = relu(weight)` the Linear uses the clamped values, and its gradient reaches the parameter
        x1 = F.linear(x, sd['class_fc_2.weight'] + (w - sd['class_fc_2.weight']).detach(), sd['class_fc_2.bias'])
        x2 = F.linear(x, sd['class_fc_22.weight'] + (w2 - sd['class_fc_22.weight']).detach(), sd['class_fc_22.bias'])
        return (x1, x2, w, w2), new
    return x, new


def _gru_last_state(sd, p, wemb, lengths, hidden):
    """embedding -> bi-GRU -> output at the last valid step (language_model.py:95-106 / caption_encoder.py:89-100)."""
    rnn = nn.GRU(wemb.shape[-1], hidden, bidirectional=True, batch_first=True)
    names = [n for n, _ in rnn.named_parameters()]
    params = {n: sd[p + n] for n in names}
    packed = pack_padded_sequence(wemb, lengths, batch_first=True)
    rnn_out, _ = torch.func.functional_call(rnn, params, (packed,))
    padded = pad_packed_sequence(rnn_out, batch_first=True)
    I = lengths.expand(2 * hidden, 1, -1).permute(2, 1, 0) - 1
    return torch.gather(padded[0], 1, I).squeeze(1)


def _pie(sd, p, out, x, mask):
    return pie_head(out, x, sd[p + 'attention.w_1.weight'], sd[p + 'attention.w_2.weight'], sd[p + 'fc.weight'],
                    sd[p + 'fc.bias'], sd[p + 'layer_norm.weight'], sd[p + 'layer_norm.bias'], pad_mask=mask)


def _head_proj(sd, p, x, train):
    """nn.Sequential(Linear(512,512), BatchNorm1d(512), ReLU, Linear(512,512)) (image_encoder.py:42-48)."""
    x = F.linear(x, sd[p + '0.weight'], sd[p + '0.bias'])
    x = F.batch_norm(x, sd[p + '1.running_mean'].clone(), sd[p + '1.running_var'].clone(), sd[p + '1.weight'],
                     sd[p + '1.bias'], training=train, momentum=0.1, eps=1e-5)
    return F.linear(F.relu(x), sd[p + '3.weight'], sd[p + '3.bias'])


def text_client_forward(sd, x, lengths, is_train, scale=128):
    """language_model.py:93-130.  Returns (outputs, new_sd_entries) like resnet_client_forward."""
    lengths = lengths.cpu()
    embed_dim = sd['class_fc.weight'].shape[1]
    wemb = F.embedding(x, sd['embed.weight'])
    out = _gru_last_state(sd, 'rnn.', wemb, lengths, embed_dim // 2)
    pad_mask = torch.arange(wemb.shape[1])[None, :] >= lengths[:, None]
    out, _, _ = _pie(sd, 'pie_net.', out, wemb, pad_mask)
    out = F.relu(out * scale)                           # :109-110
    new = {}
    if is_train:                                        # :112-121
        w = F.relu(sd['class_fc.weight'])
        w2 = F.relu(sd['class_fc_2.weight'])
        new['class_fc.weight'], new['class_fc_2.weight'] = w.detach(), w2.detach()
        x1 = F.linear(out, sd['class_fc.weight'] + (w - sd['class_fc.weight']).detach(), sd['class_fc.bias'])
        x2 = F.linear(out, sd['class_fc_2.weight'] + (w2 - sd['class_fc_2.weight']).detach(), sd['class_fc_2.bias'])
        return (x1, x2, w, w2), new
    return F.normalize(out, p=2, dim=1), new            # :126-127 (mlp_local = False)


def pcme_towers_forward(sd, fmap, sentences, lengths, mlp_local=False, train_mode=True):
    """pcme.py:35-57 (not_bert) on the trunk's output map: image_encoder.py:54-71 | caption_encoder.py:87-116.
    Returns the 10-key dict of PCME.forward."""
    n, cd = fmap.shape[0], fmap.shape[1]
    pooled = fmap.mean(dim=(2, 3)).view(-1, cd)                                        # image_encoder.py:55-56
    out = F.linear(pooled, sd['img_enc.fc.weight'], sd['img_enc.fc.bias'])
    x = fmap.view(-1, cd, 49).transpose(1, 2)
    out, _, _ = _pie(sd, 'img_enc.pie_net.', out, x, None)
    if mlp_local:
        out = _head_proj(sd, 'img_enc.head_proj.', out, train_mode)                    # :64-65: BEFORE l2norm
    img = l2_normalize(out)
    lengths = lengths.cpu()
    wemb = F.embedding(sentences, sd['txt_enc.embed.weight'])
    embed_dim = sd['txt_enc.pie_net.fc.weight'].shape[0]
    o = _gru_last_state(sd, 'txt_enc.rnn.', wemb, lengths, embed_dim // 2)
    pad_mask = torch.arange(wemb.shape[1])[None, :] >= lengths[:, None]
    o, _, _ = _pie(sd, 'txt_enc.pie_net.', o, wemb, pad_mask)
    o = l2_normalize(o)                                                                # caption_encoder.py:109
    if mlp_local:
        o = _head_proj(sd, 'txt_enc.head_proj.', o, train_mode)                        # :111-112: AFTER l2norm
    return {'image_features': img, 'image_attentions': None, 'image_residuals': None, 'image_logsigma': None,
            'image_logsigma_att': None, 'caption_features': o, 'caption_attentions': None, 'caption_residuals': None,
            'caption_logsigma': None, 'caption_logsigma_att': None}
