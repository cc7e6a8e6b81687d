#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES on seeded
inputs.  Build-container only: needs the read-only checkout at /root/reference.
The reference source never ships; only inputs + outputs are stored.

    python tests/golden/make_golden.py

What is imported from the reference:
  src/criterions/probemb.py            MCSoftContrastiveLoss         -> a1_*.npz
  src/networks/models/pie_model.py     PIENet (by file path)          -> a2_*.npz
  src/losses/__init__.py               create('softmax')              -> a34_*.npz, a34mm_*.npz (MMClientTrainer.py:164-206,:246-264,:301-308)
  src/algorithms/eval_coco.py          COCOEvaluator.evaluate_recall  -> a6_*.npz
  src/utils/tensor_utils.py            l2_normalize
  src/utils/Utils.py                   to_one_hot (+ create('softmax'))  -> f4_*.npz
  src/networks/resnet_client.py        ResNet.forward, both phases (by path)        -> a2c_img_*.npz
  src/networks/language_model.py       EncoderText.forward, both modes (by path)    -> a2c_txt_*.npz
  src/networks/models/{image_encoder,caption_encoder,pcme}.py  the reference's own EncoderImage.forward /
      EncoderText.forward / PCME.forward run on a synthetic 7x7 feature map (the torchvision trunk is replaced by
      nn.Identity; `torchvision` / `torchtext` are import-time stubs only)                      -> tower_*.npz
  MMFL.py:346-378 (KD terms; inline, literal statement sequence with nn.MSELoss as at :296)      -> kd_*.npz
  src/main.py:38-105  the argparse flag surface, extracted with ast (never imported)            -> main_flags.json
Parameters of the encoder fixtures are not stored: they are regenerated from the state_dict key names by
tests/golden/seeded.py on both sides.
Rows A3/A4/A5 are inline loop bodies in the reference (ClientTrainer.py:369-429,
MMFL.py:298-335); here their literal statement sequence is evaluated with the
imported criterion object and plain torch ops, which is what pins the oracle.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_by_path(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Cfg(dict):
    __getattr__ = dict.get


def _unit(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


def make_a1(probemb):
    for (n, d, seed, matched) in [(8, 16, 0, False), (32, 256, 0, False), (128, 256, 1, False),
                                  (48, 512, 2, True), (33, 100, 3, True)]:
        gen = torch.Generator().manual_seed(seed)
        I = _unit(gen, n, d)
        T = _unit(gen, n, d) if not matched else torch.nn.functional.normalize(
            I + 0.5 * _unit(gen, n, d), dim=-1)
        for (a0, b0) in [(15.0, 15.0), (5.0, 3.0)]:
            crit = probemb.MCSoftContrastiveLoss(
                _Cfg(init_shift=b0, init_negative_scale=a0, num_samples=7))
            Ig = I.clone().requires_grad_(True)
            Tg = T.clone().requires_grad_(True)
            loss, ld = crit(Ig, Tg, None, None)
            loss.backward()
            mp = crit.match_prob(I, T, None, None).detach()
            np.savez(os.path.join(OUT, f'a1_n{n}_d{d}_a{int(a0)}_b{int(b0)}.npz'),
                     I=I.numpy(), T=T.numpy(), a=np.float32(a0), b=np.float32(b0),
                     loss=loss.detach().numpy(),
                     dict_keys=np.array(list(ld.keys())),
                     dict_vals=np.array([float(v) for v in ld.values()], dtype=np.float64),
                     dI=Ig.grad.numpy(), dT=Tg.grad.numpy(),
                     da=crit.negative_scale.grad.numpy(), db=crit.shift.grad.numpy(),
                     match_prob=mp.numpy())


def make_a2(pie_model, tensor_utils):
    cases = [('small', 3, 10, 64, 32, 32, False, 11), ('wide', 2, 49, 1024, 64, 64, False, 12),
             ('gru_mask', 5, 12, 300, 256, 150, True, 13), ('r18', 3, 49, 512, 128, 256, False, 14),
             ('mask_vec', 4, 12, 64, 48, 40, True, 15)]      # pad mask on vector-friendly widths (single-pass kernels)
    for (tag, b, p, cd, d, dh, masked, seed) in cases:
        torch.manual_seed(seed)
        net = pie_model.PIENet(1, cd, d, dh)
        with torch.no_grad():
            net.layer_norm.weight.copy_(1.0 + 0.1 * torch.randn(d))
            net.layer_norm.bias.copy_(0.1 * torch.randn(d))
            net.fc.bias.copy_(0.05 * torch.randn(d))
        x = torch.randn(b, p, cd, requires_grad=True)
        out = torch.randn(b, d, requires_grad=True)
        mask = None
        if masked:
            lens = torch.tensor([12, 9, 7, 4, 1][:b])
            mask = torch.arange(p)[None, :] >= lens[:, None]
        o, attn, res = net(out, x, mask)
        y = tensor_utils.l2_normalize(o)
        gy = torch.randn_like(y)
        (y * gy).sum().backward()
        sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
        grads = {('g_' + k): v.grad.numpy() for k, v in net.named_parameters()}
        np.savez(os.path.join(OUT, f'a2_{tag}.npz'), x=x.detach().numpy(), out=out.detach().numpy(),
                 mask=(mask.numpy() if mask is not None else np.zeros((0,), bool)),
                 o=o.detach().numpy(), attn=attn.detach().numpy(), res=res.detach().numpy(),
                 y=y.detach().numpy(), gy=gy.numpy(), dx=x.grad.numpy(), dout=out.grad.numpy(),
                 **{('p_' + k.replace('.', '__')): v for k, v in sd.items()},
                 **{k.replace('.', '__'): v for k, v in grads.items()})


def make_a34(losses_mod):
    criterion = losses_mod.create('softmax')           # nn.CrossEntropyLoss (losses/__init__.py:19)
    for (tag, b, m, d, w, scale, seed) in [('b8_m64_d16', 8, 64, 16, 0.5, False, 0),
                                           ('b32_m1000_d256', 32, 1000, 256, 0.5, False, 1),
                                           ('b32_m1000_d256_ls', 32, 1000, 256, 0.5, True, 1),
                                           ('b128_m2000_d128', 128, 2000, 128, 0.5, False, 2),
                                           ('b17_m333_d96', 17, 333, 96, 0.25, True, 3)]:
        gen = torch.Generator().manual_seed(seed)
        g_img = _unit(gen, m, d)
        g_txt = torch.nn.functional.normalize(g_img + 0.7 * _unit(gen, m, d), dim=-1)
        d_idx = tuple(int(v) for v in torch.randperm(m, generator=gen)[:b])
        base = g_img[list(d_idx)]
        im_feature = torch.nn.functional.normalize(base + 0.6 * _unit(gen, b, d), dim=-1).requires_grad_(True)
        old_im_feature = torch.nn.functional.normalize(base + 0.6 * _unit(gen, b, d), dim=-1)
        # ---- literal statement sequence of ClientTrainer.py:386-419 (image client) ----
        target_feature = g_img[d_idx, :].type_as(im_feature)
        logits_inter = torch.div(torch.matmul(im_feature, g_txt.T), 0.5)
        labels_inter = torch.tensor(d_idx)
        loss_inter = criterion(logits_inter, labels_inter)
        pos = torch.sum(im_feature * target_feature, dim=-1)
        pos = pos.reshape(-1, 1)
        neg = torch.sum(im_feature * old_im_feature, dim=-1)
        logits = torch.cat((pos, neg.reshape(-1, 1)), dim=1)
        logits = logits / 0.5
        labels = torch.zeros(b).long()
        loss_moon = criterion(logits, labels)
        if not scale:
            loss = (loss_moon + loss_inter) * w
        else:
            loss = (loss_moon + loss_inter / (loss_inter / loss_moon).detach()) * w
        loss.backward()
        # inter-only / intra-only grads (ClientTrainer.py:470, :502)
        f2 = im_feature.detach().clone().requires_grad_(True)
        li = criterion(torch.div(torch.matmul(f2, g_txt.T), 0.5), labels_inter)
        li.backward()
        f3 = im_feature.detach().clone().requires_grad_(True)
        p3 = torch.sum(f3 * target_feature, dim=-1).reshape(-1, 1)
        n3 = torch.sum(f3 * old_im_feature, dim=-1)
        lm = criterion(torch.cat((p3, n3.reshape(-1, 1)), dim=1) / 0.5, labels)
        lm.backward()
        np.savez(os.path.join(OUT, f'a34_{tag}.npz'), f=im_feature.detach().numpy(),
                 f_old=old_im_feature.numpy(), g_same=g_img.numpy(), g_other=g_txt.numpy(),
                 d_idx=np.array(d_idx, dtype=np.int64), weight=np.float32(w),
                 loss_scale=np.bool_(scale), loss=loss.detach().numpy(),
                 loss_inter=loss_inter.detach().numpy(), loss_moon=loss_moon.detach().numpy(),
                 df=im_feature.grad.numpy(), df_inter_only=f2.grad.numpy(),
                 df_intra_only=f3.grad.numpy())


def make_a34_mm(losses_mod):
    """Multi-modal client (MMClientTrainer.py:150-324): the literal statement sequences of the three flag branches
    (:164-206 both, :246-264 intra only, :301-308 inter only) on CPU tensors (the reference's `.cuda()` calls dropped),
    with the imported criterion object (`nn.CrossEntropyLoss().cuda()` at :149 is the same class
    losses.create('softmax') returns)."""
    criterion = losses_mod.create('softmax')
    for (tag, b, m, d, w, scale, seed, dup) in [('b8_m64_d16', 8, 64, 16, 0.5, False, 10, False),
                                                ('b32_m1000_d256', 32, 1000, 256, 0.5, False, 11, False),
                                                ('b32_m1000_d256_ls', 32, 1000, 256, 0.5, True, 11, False),
                                                ('b19_m333_d96_ls', 19, 333, 96, 0.25, True, 12, False),
                                                ('b21_m500_d128_dup', 21, 500, 128, 0.5, False, 13, True),
                                                ('b64_m1200_d512', 64, 1200, 512, 0.5, False, 14, False)]:
        gen = torch.Generator().manual_seed(seed)
        global_img_feature = _unit(gen, m, d)
        global_txt_feature = torch.nn.functional.normalize(global_img_feature + 0.7 * _unit(gen, m, d), dim=-1)
        perm = [int(v) for v in torch.randperm(m, generator=gen)[:b]]
        if dup:                                     # repeated bank positions inside one batch
            perm[3] = perm[0]
            perm[-1] = perm[5]
        d_idx = tuple(perm)
        base_i = global_img_feature[list(d_idx)]
        base_t = global_txt_feature[list(d_idx)]
        out_img = torch.nn.functional.normalize(base_i + 0.6 * _unit(gen, b, d), dim=-1).requires_grad_(True)
        out_txt = torch.nn.functional.normalize(base_t + 0.6 * _unit(gen, b, d), dim=-1).requires_grad_(True)
        out_img_o = torch.nn.functional.normalize(base_i + 0.6 * _unit(gen, b, d), dim=-1)
        out_txt_o = torch.nn.functional.normalize(base_t + 0.6 * _unit(gen, b, d), dim=-1)
        images = out_img                              # only `images.size(0)` is used below
        # ---- MMClientTrainer.py:164-206 (both flags) ----
        target_img_feature = global_img_feature[d_idx, :].type_as(out_img)
        target_txt_feature = global_txt_feature[d_idx, :].type_as(out_txt)
        pos_i = torch.sum(out_img * target_img_feature, dim=-1)
        pos_i = pos_i.reshape(-1, 1)
        pos_t = torch.sum(out_txt * target_txt_feature, dim=-1)
        pos_t = pos_t.reshape(-1, 1)
        neg_i = torch.sum(out_img * out_img_o, dim=-1)
        neg_t = torch.sum(out_txt * out_txt_o, dim=-1)
        logits_1 = torch.cat((pos_i, neg_i.reshape(-1, 1)), dim=1)
        logits_2 = torch.cat((pos_t, neg_t.reshape(-1, 1)), dim=1)
        logits = torch.cat((logits_1, logits_2), dim=0)
        logits /= 0.5  # temperature
        labels = torch.zeros(images.size(0) * 2).long()
        loss_intra = criterion(logits, labels)
        logits_1_inter = torch.div(torch.matmul(out_img, global_txt_feature.T), 0.5)
        logits_2_inter = torch.div(torch.matmul(out_txt, global_img_feature.T), 0.5)
        labels_inter = torch.tensor(d_idx)
        loss_1_inter = criterion(logits_1_inter, labels_inter)
        loss_2_inter = criterion(logits_2_inter, labels_inter)
        loss_inter = loss_1_inter + loss_2_inter
        if not scale:
            loss = (loss_intra + loss_inter) * w
        else:
            loss = (loss_intra + loss_inter / (loss_inter / loss_intra).detach()) * w
        loss.backward()
        # ---- :246-264 (intra only; unweighted) ----
        i2 = out_img.detach().clone().requires_grad_(True)
        t2 = out_txt.detach().clone().requires_grad_(True)
        p_i = torch.sum(i2 * target_img_feature, dim=-1).reshape(-1, 1)
        p_t = torch.sum(t2 * target_txt_feature, dim=-1).reshape(-1, 1)
        n_i = torch.sum(i2 * out_img_o, dim=-1)
        n_t = torch.sum(t2 * out_txt_o, dim=-1)
        lg = torch.cat((torch.cat((p_i, n_i.reshape(-1, 1)), dim=1), torch.cat((p_t, n_t.reshape(-1, 1)), dim=1)), dim=0)
        lg /= 0.5
        loss_intra_only = criterion(lg, labels)
        loss_intra_only.backward()
        # ---- :301-308 (inter only; unweighted) ----
        i3 = out_img.detach().clone().requires_grad_(True)
        t3 = out_txt.detach().clone().requires_grad_(True)
        l1 = criterion(torch.div(torch.matmul(i3, global_txt_feature.T), 0.5), labels_inter)
        l2 = criterion(torch.div(torch.matmul(t3, global_img_feature.T), 0.5), labels_inter)
        loss_inter_only = l1 + l2
        loss_inter_only.backward()
        np.savez(os.path.join(OUT, f'a34mm_{tag}.npz'), out_img=out_img.detach().numpy(), out_txt=out_txt.detach().numpy(),
                 old_img=out_img_o.numpy(), old_txt=out_txt_o.numpy(), g_img=global_img_feature.numpy(),
                 g_txt=global_txt_feature.numpy(), d_idx=np.array(d_idx, dtype=np.int64), weight=np.float32(w),
                 loss_scale=np.bool_(scale), loss=loss.detach().numpy(), loss_inter=loss_inter.detach().numpy(),
                 loss_intra=loss_intra.detach().numpy(), d_img=out_img.grad.numpy(), d_txt=out_txt.grad.numpy(),
                 loss_intra_only=loss_intra_only.detach().numpy(), d_img_intra_only=i2.grad.numpy(),
                 d_txt_intra_only=t2.grad.numpy(), loss_inter_only=loss_inter_only.detach().numpy(),
                 d_img_inter_only=i3.grad.numpy(), d_txt_inter_only=t3.grad.numpy())


def make_a5():
    for (tag, m, d, c, seed) in [('m512_d64_c3', 512, 64, 3, 0), ('m600_d128_c4', 600, 128, 4, 1),
                                 ('m257_d48_c1', 257, 48, 1, 2)]:
        gen = torch.Generator().manual_seed(seed)
        g_txt = _unit(gen, m, d)
        vecs = [torch.nn.functional.normalize(g_txt + (0.3 + 0.4 * i) * _unit(gen, m, d), dim=-1)
                for i in range(c)]
        # ---- literal statement sequence of MMFL.py:300-314 ----
        i_vec = [v.clone() for v in vecs]
        num_i_vec = len(i_vec)
        contrastive_w = torch.zeros(num_i_vec, m)
        for i_idx, vec in enumerate(i_vec):
            logits = torch.matmul(vec, g_txt.T)
            exp_logits = torch.exp(logits)
            log_prob = logits - torch.log(torch.sum(exp_logits, dim=1, keepdim=True))
            contrastive_w[i_idx] = torch.diagonal(log_prob).reshape(-1)
        logprob = contrastive_w.clone()
        contrastive_w[:num_i_vec] = torch.softmax(contrastive_w[:num_i_vec], dim=0)
        for i in range(len(i_vec)):
            i_vec[i] = (i_vec[i] * contrastive_w[i].reshape(-1, 1)).unsqueeze(0)
        agg = torch.sum(torch.cat(i_vec, dim=0), dim=0)
        np.savez(os.path.join(OUT, f'a5_{tag}.npz'), vecs=torch.stack(vecs).numpy(),
                 g_other=g_txt.numpy(), logprob=logprob.numpy(), weights=contrastive_w.numpy(),
                 agg=agg.numpy())


def make_a6(eval_coco):
    for (tag, n_img, cap_per, d, noise, seed) in [('i40_d32', 40, 5, 32, 0.8, 0), ('i200_d64', 200, 5, 64, 1.2, 1),
                                                  ('i400_d128', 400, 5, 128, 1.5, 2)]:
        gen = torch.Generator().manual_seed(seed)
        img = _unit(gen, n_img, d)
        cap = torch.nn.functional.normalize(
            img.repeat_interleave(cap_per, 0) + noise * _unit(gen, n_img * cap_per, d), dim=-1)
        img_cls = np.arange(n_img)
        cap_cls = np.arange(n_img * cap_per) // cap_per
        ev = eval_coco.COCOEvaluator(eval_method='matmul', verbose=False, eval_device='cpu', n_crossfolds=5)
        ev.n_embeddings = 7
        # extract_features :135-136,175,181 -> fp64 [n, 7, D] buffers with 7 identical copies
        imgf = torch.from_numpy(np.repeat(img.numpy().astype(np.float64)[:, None, :], 7, 1).copy())
        capf = torch.from_numpy(np.repeat(cap.numpy().astype(np.float64)[:, None, :], 7, 1).copy())
        i2t = ev.evaluate_recall(imgf, capf, img_cls, cap_cls, batch_size=64)
        t2i = ev.evaluate_recall(capf, imgf, cap_cls, img_cls, batch_size=64)
        keys = ['recall_1', 'recall_5', 'recall_10', 'rsum', 'medr', 'meanr']
        np.savez(os.path.join(OUT, f'a6_{tag}.npz'), img=img.numpy(), cap=cap.numpy(),
                 img_cls=img_cls, cap_cls=cap_cls, keys=np.array(keys),
                 i2t=np.array([i2t[k] for k in keys], dtype=np.float64),
                 t2i=np.array([t2i[k] for k in keys], dtype=np.float64))


def make_f4(losses_mod, utils_mod):
    """literal statement sequence of ClientTrainer.py:344-357 with the reference's own to_one_hot
    and criterion objects (the ClientTrainer module itself needs apex/torchvision and cannot be imported)."""
    criterion = losses_mod.create('softmax')
    for (tag, b, c, dw, margin, k5, seed) in [('cifar100', 64, 100, 512, 4.0, 5, 0), ('cifar10', 50, 10, 512, 4.0, 5, 1),
                                              ('agnews', 33, 4, 256, 4.0, 4, 2), ('yelp', 16, 2, 256, 4.0, 2, 3),
                                              ('wide', 7, 300, 96, 1.5, 5, 4)]:
        gen = torch.Generator().manual_seed(100 + seed)
        labels_var = torch.randint(0, c, (b,), generator=gen)
        class_weight = torch.relu(torch.randn(c, dw, generator=gen) * 0.05).requires_grad_(True)
        fvec0 = (torch.randn(b, c, generator=gen) * 2.0)
        fvec0[torch.arange(b), labels_var] += 3.0               # mostly-right classifier: precision is not trivial
        fvec0.requires_grad_(True)
        class_label = torch.Tensor(np.array(range(c)))          # ClientTrainer.py:266
        center_labels_var = class_label.to(torch.long)
        labels_var_one_hot = utils_mod.to_one_hot(labels_var, n_dims=c)
        fvec = fvec0 - margin * labels_var_one_hot
        loss = criterion(fvec, labels_var)
        center_loss = criterion(torch.mm(class_weight, torch.t(class_weight)), center_labels_var)
        total_loss = 0.5 * center_loss + loss
        # accuracy(fvec.data, labels_bt, topk=(1, k5))  (ClientTrainer.py:114-129; `.to(gpuid)` is the identity on CPU)
        maxk = max((1, k5))
        _, pred = fvec.data.topk(maxk, 1, True, True)
        pred = pred.t()
        correct = pred.eq(labels_var.view(1, -1).expand_as(pred))
        prec = [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / b) for k in (1, k5)]
        total_loss.backward()
        np.savez(os.path.join(OUT, f'f4_{tag}.npz'), fvec=fvec0.detach().numpy(), labels=labels_var.numpy(),
                 class_weight=class_weight.detach().numpy(), margin=np.float32(margin), topk=np.int64(k5),
                 total=total_loss.detach().numpy(), ce=loss.detach().numpy(), center=center_loss.detach().numpy(),
                 prec1=prec[0].numpy(), preck=prec[1].numpy(), dfvec=fvec0.grad.numpy(),
                 dclass_weight=class_weight.grad.numpy())


def _stub_vision_text():
    """Import-time stand-ins for the two absent packages.  Nothing of them is CALLED: resnet_client.py:7 imports an
    unused name, image_encoder.py's torchvision trunk is replaced by nn.Identity below, GloVe is skipped by
    wemb_type=None.  transformers must be imported first (it probes torchvision's import spec)."""
    import transformers  # noqa: F401
    from transformers import BertModel, BertTokenizer  # noqa: F401
    for n in ['torchvision', 'torchvision.models', 'torchtext']:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision.models'].resnet18 = None


def _grad_dict(model, names):
    named = dict(model.named_parameters())
    return {('g_' + n.replace('.', '__')): named[n].grad.detach().numpy().copy() for n in names}


def make_a2c_img(resnet_client):
    """resnet_client.ResNet.forward :175-201: phase 'extract_conv_feature' (l2-normalised embedding) and the classifier
    phase with its `weight.data = relu(weight)` side effect, in train mode (batch statistics) and in eval mode."""
    from seeded import seeded_state_dict, checksum
    for (tag, d, train, seed) in [('d64_train', 64, True, 21), ('d512_eval', 512, False, 22)]:
        model = resnet_client.ResNet(resnet_client.BasicBlock, [1, 1, 1, 1], embed_dim=d, num_class=10, is_train=True,
                                     scale=128, phase='none')
        sd = seeded_state_dict(model.state_dict(), seed)
        model.load_state_dict(sd)
        model.train(train)
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(4, 3, 64, 64, generator=gen)
        gnames = ['conv1.weight', 'bn1.weight', 'layer4.0.bn2.bias', 'layer2.0.downsample.0.weight'] + (['linear.weight'] if d != 512 else [])
        # --- ClientTrainer.py:372-375: model.phase = 'extract_conv_feature'
        model.phase = 'extract_conv_feature'
        feat = model(x)
        gy = torch.randn(feat.shape, generator=gen)
        model.zero_grad()
        (feat * gy).sum().backward()
        g_feat = _grad_dict(model, gnames)
        rm_after = model.bn1.running_mean.detach().numpy().copy()
        # --- supervised phase (ClientTrainer.py:344): returns (x1, x2, relu(W), relu(W2)) and clamps the weights
        model.phase = 'none'
        model.zero_grad()
        x1, x2, w, w2 = model(x)
        g1 = torch.randn(x1.shape, generator=gen)
        g2 = torch.randn(x2.shape, generator=gen)
        ((x1 * g1).sum() + (x2 * g2).sum() + 0.1 * (w ** 2).sum()).backward()
        g_cls = _grad_dict(model, ['class_fc_2.weight', 'class_fc_2.bias', 'class_fc_22.weight', 'conv1.weight'])
        np.savez(os.path.join(OUT, f'a2c_img_{tag}.npz'), seed=np.int64(seed), embed_dim=np.int64(d), train=np.bool_(train),
                 wsum=np.float64(checksum(sd)), x=x.numpy(), feat=feat.detach().numpy(), gy=gy.numpy(),
                 bn1_running_mean_after_feat=rm_after, x1=x1.detach().numpy(), x2=x2.detach().numpy(),
                 w=w.detach().numpy(), w2=w2.detach().numpy(), g1=g1.numpy(), g2=g2.numpy(),
                 class_fc_2_weight_after=model.class_fc_2.weight.detach().numpy(),
                 class_fc_22_weight_after=model.class_fc_22.weight.detach().numpy(),
                 **{('feat__' + k): v for k, v in g_feat.items()}, **{('cls__' + k): v for k, v in g_cls.items()})


def make_a2c_txt(language_model):
    """language_model.EncoderText.forward :93-130: is_train heads (with the ReLU weight clamp) and the l2norm path."""
    from seeded import seeded_state_dict, checksum
    cwd = os.getcwd()
    os.chdir(REF)                                              # opens src/datasets/vocabs/coco_vocab.pkl (:31)
    try:
        for (tag, d, seed) in [('d64', 64, 31), ('d256', 256, 32)]:
            model = language_model.EncoderText(wemb_type=None, word_dim=300, embed_dim=d, num_class=4, scale=128)
            sd = seeded_state_dict(model.state_dict(), seed)
            model.load_state_dict(sd)
            model.train()
            gen = torch.Generator().manual_seed(seed)
            lengths = torch.tensor([12, 9, 7, 4, 1])
            x = torch.randint(4, model.embed.weight.shape[0], (5, 12), generator=gen)
            x = x * (torch.arange(12)[None, :] < lengths[:, None])       # 0 = <pad>
            gnames = ['pie_net.fc.weight', 'pie_net.attention.w_1.weight', 'pie_net.layer_norm.weight', 'rnn.weight_hh_l0']
            model.is_train = False
            feat = model(x, lengths)
            gy = torch.randn(feat.shape, generator=gen)
            model.zero_grad()
            (feat * gy).sum().backward()
            g_feat = _grad_dict(model, gnames)
            model.is_train = True
            model.zero_grad()
            x1, x2, w, w2 = model(x, lengths)
            g1 = torch.randn(x1.shape, generator=gen)
            g2 = torch.randn(x2.shape, generator=gen)
            ((x1 * g1).sum() + (x2 * g2).sum() + 0.1 * (w ** 2).sum()).backward()
            g_cls = _grad_dict(model, ['class_fc.weight', 'class_fc.bias', 'class_fc_2.weight', 'pie_net.fc.weight'])
            np.savez(os.path.join(OUT, f'a2c_txt_{tag}.npz'), seed=np.int64(seed), embed_dim=np.int64(d),
                     vocab=np.int64(model.embed.weight.shape[0]), wsum=np.float64(checksum(sd)), x=x.numpy(),
                     lengths=lengths.numpy(), feat=feat.detach().numpy(), gy=gy.numpy(), x1=x1.detach().numpy(),
                     x2=x2.detach().numpy(), w=w.detach().numpy(), w2=w2.detach().numpy(), g1=g1.numpy(), g2=g2.numpy(),
                     class_fc_weight_after=model.class_fc.weight.detach().numpy(),
                     **{('feat__' + k): v for k, v in g_feat.items()}, **{('cls__' + k): v for k, v in g_cls.items()})
    finally:
        os.chdir(cwd)


def make_tower(pcme_mod, image_encoder, caption_encoder):
    """The reference's own PCME.forward (pcme.py:35-57) -> EncoderImage.forward (image_encoder.py:54-71) +
    EncoderText.forward (caption_encoder.py:87-116) on a synthetic [N, Cd, 7, 7] trunk output (cnn = nn.Identity).
    'mlp' exercises head_proj (hard-wired 512, image_encoder.py:42-48) and the caption tower's l2norm-BEFORE-head_proj
    order (caption_encoder.py:109-112)."""
    import torch.nn as nn
    from seeded import seeded_state_dict, checksum
    for (tag, cd, d, mlp, seed) in [('d32', 64, 32, False, 41), ('d512_mlp', 128, 512, True, 42), ('r18_d256', 512, 256, False, 43)]:
        cfg = _Cfg(embed_dim=d, wemb_type=None, word_dim=300, cache_dir=None, not_bert=True, n_samples_inference=7,
                   cnn_type='resnet18')
        word2idx = {str(i): i for i in range(60)}
        txt = caption_encoder.EncoderText(word2idx, cfg, mlp)
        img = image_encoder.EncoderImage.__new__(image_encoder.EncoderImage)
        nn.Module.__init__(img)
        img.cnn = nn.Identity()                                # the torchvision trunk: outside the glue under test
        img.cnn_dim = cd
        img.avgpool = nn.AdaptiveAvgPool2d((1, 1))             # what torchvision's resnet.avgpool is
        img.fc = nn.Linear(cd, d)
        img.pie_net = image_encoder.PIENet(1, cd, d, cd // 2)
        img.mlp_local = mlp
        if mlp:
            img.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Linear(512, 512))
        model = pcme_mod.PCME.__new__(pcme_mod.PCME)
        nn.Module.__init__(model)
        model.config, model.embed_dim, model.n_embeddings = cfg, d, 7
        model.img_enc, model.txt_enc = img, txt
        sd = seeded_state_dict(model.state_dict(), seed)
        model.load_state_dict(sd)
        model.train()
        gen = torch.Generator().manual_seed(seed)
        n = 6
        fmap = torch.randn(n, cd, 7, 7, generator=gen).requires_grad_(True)
        lengths = torch.tensor([9, 8, 6, 5, 3, 2])
        sent = torch.randint(1, 60, (n, 9), generator=gen) * (torch.arange(9)[None, :] < lengths[:, None])
        out = model(fmap, sent, None, lengths)
        keys = list(out.keys())
        gi = torch.randn(n, d, generator=gen)
        gc = torch.randn(n, d, generator=gen)
        ((out['image_features'] * gi).sum() + (out['caption_features'] * gc).sum()).backward()
        gn = ['img_enc.fc.weight', 'img_enc.pie_net.attention.w_1.weight', 'img_enc.pie_net.layer_norm.bias',
              'txt_enc.pie_net.fc.weight', 'txt_enc.rnn.weight_ih_l0'] + (['img_enc.head_proj.0.weight', 'txt_enc.head_proj.3.weight'] if mlp else [])
        np.savez(os.path.join(OUT, f'tower_{tag}.npz'), seed=np.int64(seed), cd=np.int64(cd), embed_dim=np.int64(d),
                 mlp_local=np.bool_(mlp), wsum=np.float64(checksum(sd)), fmap=fmap.detach().numpy(), sentences=sent.numpy(),
                 lengths=lengths.numpy(), keys=np.array(keys), none_keys=np.array([k for k in keys if out[k] is None]),
                 image_features=out['image_features'].detach().numpy(), caption_features=out['caption_features'].detach().numpy(),
                 gi=gi.numpy(), gc=gc.numpy(), dfmap=fmap.grad.numpy(), **_grad_dict(model, gn))


def make_kd():
    """MMFL.py:346-378, literal statement sequence of the KD terms with client_loss_cri = nn.MSELoss() (:296):
    one `kd_weight * code_sim` per client type -- the image term is added TWICE when image and multimodal clients
    both exist (:361-378)."""
    import operator
    import torch.nn as nn
    client_loss_cri = nn.MSELoss()
    for (tag, m, b, d, n_img, n_txt, n_mm, kd_weight, three_d, seed) in [
            ('all_types', 300, 16, 64, 2, 2, 1, 0.3, False, 51), ('img_only', 200, 8, 32, 3, 0, 0, 1.0, False, 52),
            ('txt_mm', 257, 5, 48, 0, 1, 2, 0.5, False, 53), ('three_d', 128, 6, 16, 1, 1, 0, 0.3, True, 54)]:
        gen = torch.Generator().manual_seed(seed)
        img_vec = _unit(gen, m, d)
        txt_vec = _unit(gen, m, d)
        distill_index = [int(v) for v in torch.randperm(10 * m, generator=gen)[:m]]     # coco ids of the public set
        distill_dict = {b_: a for a, b_ in enumerate(distill_index)}                   # MMFL.py:342
        index = [distill_index[int(i)] for i in torch.randperm(m, generator=gen)[:b]]   # ids of this batch
        shape = (b, 7, d) if three_d else (b, d)
        oi = torch.randn(*shape, generator=gen)
        ot = torch.randn(*shape, generator=gen)
        if not three_d:
            oi, ot = torch.nn.functional.normalize(oi, dim=-1), torch.nn.functional.normalize(ot, dim=-1)
        oi.requires_grad_(True)
        ot.requires_grad_(True)
        output = {'image_features': oi, 'caption_features': ot}
        loss = 0

        def code_sim(output, target, config):
            output = output.sum(axis=1) if len(output.shape) == 3 else output
            target = target.type_as(output)
            return client_loss_cri(output, target.type_as(output))

        if n_img > 0:
            out_img = output['image_features']
            d_idx = operator.itemgetter(*index)(distill_dict)
            target_img = img_vec[d_idx, :].type_as(out_img)
            loss += kd_weight * code_sim(out_img, target_img, None)
        if n_txt > 0:
            out_txt = output['caption_features']
            d_idx = operator.itemgetter(*index)(distill_dict)
            target_txt = txt_vec[d_idx, :].type_as(out_txt)
            loss += kd_weight * code_sim(out_txt, target_txt, None)
        if n_mm > 0:
            out_img = output['image_features']
            d_idx = operator.itemgetter(*index)(distill_dict)
            target_img = img_vec[d_idx, :].type_as(out_img)
            out_txt = output['caption_features']
            target_txt = txt_vec[d_idx, :].type_as(out_txt)
            loss += kd_weight * code_sim(out_img, target_img, None)
            loss += kd_weight * code_sim(out_txt, target_txt, None)
        loss.backward()
        np.savez(os.path.join(OUT, f'kd_{tag}.npz'), img_vec=img_vec.numpy(), txt_vec=txt_vec.numpy(),
                 distill_index=np.array(distill_index, dtype=np.int64), index=np.array(index, dtype=np.int64),
                 d_idx=np.array(d_idx, dtype=np.int64), out_img=oi.detach().numpy(), out_txt=ot.detach().numpy(),
                 num_img_clients=np.int64(n_img), num_txt_clients=np.int64(n_txt), num_mm_clients=np.int64(n_mm),
                 kd_weight=np.float32(kd_weight), loss=loss.detach().numpy(),
                 d_out_img=(oi.grad.numpy() if oi.grad is not None else np.zeros(shape, np.float32)),
                 d_out_txt=(ot.grad.numpy() if ot.grad is not None else np.zeros(shape, np.float32)))


def make_main_flags():
    """The flag surface of src/main.py:38-105 as DATA: every `parser.add_argument` of its `args()` function -> option
    strings, dest, type name, action and default (ast only; main.py itself is never imported: it parses sys.argv at import
    and pulls in apex / torchvision).  Defaults that are expressions (a random seed, a path under $HOME) are recorded as
    null with `computed: true`."""
    import ast
    import json
    src = open(os.path.join(REF, 'src', 'main.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'args'][0]
    flags = []
    for node in ast.walk(fn):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'add_argument'):
            continue
        opts = [a.value for a in node.args if isinstance(a, ast.Constant)]
        kw = {k.arg: k.value for k in node.keywords}
        rec = {'options': opts, 'dest': opts[-1].lstrip('-').replace('-', '_'), 'line': node.lineno}
        if 'type' in kw and isinstance(kw['type'], ast.Name):
            rec['type'] = kw['type'].id
        if 'action' in kw:
            rec['action'] = ast.literal_eval(kw['action'])
        if 'nargs' in kw:
            rec['nargs'] = ast.literal_eval(kw['nargs'])
        if 'choices' in kw:
            rec['choices'] = ast.literal_eval(kw['choices'])
        if 'default' in kw:
            try:
                rec['default'] = ast.literal_eval(kw['default'])
            except ValueError:
                rec['default'], rec['computed'] = None, True
        flags.append(rec)
    flags.sort(key=lambda r: r['line'])
    # what main.py adds to the namespace / touches on the algorithm object after parsing (main.py:112-134)
    extra = {'namespace_added_after_construction': ['save_dirs', 'log_dir'],
             'algo_surface': ['create_model', 'load_dataset', 'train', 'logger', 'engine', 'best_scores', 'best_metadata'],
             'engine_surface': ['report_scores', 'eval_prefix']}
    with open(os.path.join(OUT, 'main_flags.json'), 'w') as f:
        json.dump({'flags': flags, **extra}, f, indent=1, sort_keys=True)


def main():
    assert os.path.isdir(REF), 'reference checkout not present (build container only)'
    sys.path[:0] = [REF, os.path.join(REF, 'src')]
    torch.set_num_threads(8)
    import src.criterions.probemb as probemb
    import src.utils.tensor_utils as tensor_utils
    # src/losses/__init__.py imports its (dead) sibling loss files; they import cleanly on torch 2.x
    import src.losses as losses_mod
    pie_model = _load_by_path('pie_model', 'src/networks/models/pie_model.py')
    # eval_coco does `from src.utils.tensor_utils import to_numpy` (importable) and tqdm
    eval_coco = _load_by_path('ref_eval_coco', 'src/algorithms/eval_coco.py')
    if '--only-a2' in sys.argv:                               # one family (the others are not rewritten)
        make_a2(pie_model, tensor_utils)
        return
    if '--only-a34mm' in sys.argv:
        make_a34_mm(losses_mod)
        return
    make_a1(probemb)
    make_a2(pie_model, tensor_utils)
    make_a34(losses_mod)
    make_a34_mm(losses_mod)
    make_a5()
    make_a6(eval_coco)
    import src.utils.Utils as utils_mod
    make_f4(losses_mod, utils_mod)
    sys.path.insert(0, OUT)                                   # seeded.py
    _stub_vision_text()
    resnet_client = _load_by_path('ref_resnet_client', 'src/networks/resnet_client.py')
    language_model = _load_by_path('ref_language_model', 'src/networks/language_model.py')   # finds `pie_model` above
    make_a2c_img(resnet_client)
    make_a2c_txt(language_model)
    import src.networks.models.pcme as pcme_mod
    import src.networks.models.image_encoder as image_encoder
    import src.networks.models.caption_encoder as caption_encoder
    make_tower(pcme_mod, image_encoder, caption_encoder)
    make_kd()
    make_main_flags()
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
