"""Deterministic parameter values from (key name, shape, seed) alone.

The client / tower encoders have millions of parameters (the 11 755 x 300 word embedding, the ResNet convolutions):
far too many to store in a fixture.  Instead both sides -- make_golden.py, which loads the values into the REFERENCE's
modules, and the tests, which load them into the oracle / the product modules -- regenerate them from the state_dict
key names with this function; the fixture only holds inputs, outputs and a checksum of the weights.
"""
import zlib

import torch


def seeded_state_dict(template, seed):
    """template: {key: tensor or shape} (a module's state_dict()).  Returns {key: fp32 / int64 tensor}.
    Scales are chosen so that activations stay O(1) through the networks (fan-in scaled weights, BatchNorm
    statistics near (0, 1)) and every branch of the forward (ReLU clamps on classifier weights, LayerNorm affine,
    biases) is exercised with non-trivial values."""
    out = {}
    for key in sorted(template):
        v = template[key]
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        gen = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == 'running_var':
            t = 0.5 + torch.rand(shape, generator=gen)
        elif leaf == 'running_mean':
            t = 0.1 * torch.randn(shape, generator=gen)
        elif len(shape) <= 1 and leaf == 'weight':            # BatchNorm / LayerNorm scale
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
        elif len(shape) <= 1:                                 # biases
            t = 0.05 * torch.randn(shape, generator=gen)
        elif 'embed' in key and len(shape) == 2:              # word embedding table
            t = 0.3 * torch.randn(shape, generator=gen)
        else:                                                 # conv / linear / GRU weights: fan-in scaling
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=gen) * (1.0 / max(fan_in, 1)) ** 0.5
        out[key] = t
    return out


def checksum(sd):
    """Order-independent fp64 fingerprint of a state dict (stored in the fixture, re-checked by the tests)."""
    return float(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.is_floating_point()))


# ----------------------------------------------------------------- state_dict templates (key -> shape)
def resnet_client_template(d, num_class=10, layers=(1, 1, 1, 1)):
    """state_dict key -> shape of resnet_client.ResNet(BasicBlock, layers, embed_dim=d) (resnet_client.py:102-140)."""
    t = {'conv1.weight': (64, 3, 7, 7)}

    def bn(p, c):
        t.update({p + '.weight': (c,), p + '.bias': (c,), p + '.running_mean': (c,), p + '.running_var': (c,),
                  p + '.num_batches_tracked': ()})
    bn('bn1', 64)
    inpl = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for b in range(n):
            p = f'layer{li}.{b}'
            stride = 2 if (li > 1 and b == 0) else 1
            t[p + '.conv1.weight'] = (planes, inpl, 3, 3)
            bn(p + '.bn1', planes)
            t[p + '.conv2.weight'] = (planes, planes, 3, 3)
            bn(p + '.bn2', planes)
            if stride != 1 or inpl != planes:
                t[p + '.downsample.0.weight'] = (planes, inpl, 1, 1)
                bn(p + '.downsample.1', planes)
            inpl = planes
    if d != 512:
        t.update({'linear.weight': (d, 512), 'linear.bias': (d,)})
    t.update({'class_fc_2.weight': (num_class, d), 'class_fc_2.bias': (num_class,), 'class_fc_22.weight': (80, d),
              'class_fc_22.bias': (80,)})
    return t


def text_client_template(d, vocab, num_class=4, word_dim=300):
    t = {'embed.weight': (vocab, word_dim)}
    for sfx in ('', '_reverse'):
        t.update({f'rnn.weight_ih_l0{sfx}': (3 * (d // 2), word_dim), f'rnn.weight_hh_l0{sfx}': (3 * (d // 2), d // 2),
                  f'rnn.bias_ih_l0{sfx}': (3 * (d // 2),), f'rnn.bias_hh_l0{sfx}': (3 * (d // 2),)})
    t.update(pie_template('pie_net.', word_dim, d, word_dim // 2))
    t.update({'class_fc.weight': (num_class, d), 'class_fc.bias': (num_class,), 'class_fc_2.weight': (80, d),
              'class_fc_2.bias': (80,)})
    return t


def pie_template(p, d_in, d_out, d_h):
    return {p + 'attention.w_1.weight': (d_h, d_in), p + 'attention.w_2.weight': (1, d_h), p + 'fc.weight': (d_out, d_in),
            p + 'fc.bias': (d_out,), p + 'layer_norm.weight': (d_out,), p + 'layer_norm.bias': (d_out,)}


def head_proj_template(p):
    return {p + '0.weight': (512, 512), p + '0.bias': (512,), p + '1.weight': (512,), p + '1.bias': (512,),
            p + '1.running_mean': (512,), p + '1.running_var': (512,), p + '1.num_batches_tracked': (),
            p + '3.weight': (512, 512), p + '3.bias': (512,)}


def tower_template(cd, d, mlp, vocab=60, word_dim=300):
    t = {'img_enc.fc.weight': (d, cd), 'img_enc.fc.bias': (d,)}
    t.update(pie_template('img_enc.pie_net.', cd, d, cd // 2))
    t['txt_enc.embed.weight'] = (vocab, word_dim)
    for sfx in ('', '_reverse'):
        t.update({f'txt_enc.rnn.weight_ih_l0{sfx}': (3 * (d // 2), word_dim), f'txt_enc.rnn.weight_hh_l0{sfx}': (3 * (d // 2), d // 2),
                  f'txt_enc.rnn.bias_ih_l0{sfx}': (3 * (d // 2),), f'txt_enc.rnn.bias_hh_l0{sfx}': (3 * (d // 2),)})
    t.update(pie_template('txt_enc.pie_net.', word_dim, d, word_dim // 2))
    if mlp:
        t.update(head_proj_template('img_enc.head_proj.'))
        t.update(head_proj_template('txt_enc.head_proj.'))
    return t

