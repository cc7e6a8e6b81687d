#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES on seeded
inputs.  Build-container only: needs the read-only checkout at /root/reference.
The reference source never ships; only inputs + outputs are stored.

    python tests/golden/make_golden.py

What is imported from the reference:
  src/criterions/probemb.py            MCSoftContrastiveLoss         -> a1_*.npz
  src/networks/models/pie_model.py     PIENet (by file path)          -> a2_*.npz
  src/losses/__init__.py               create('softmax')              -> a34_*.npz
  src/algorithms/eval_coco.py          COCOEvaluator.evaluate_recall  -> a6_*.npz
  src/utils/tensor_utils.py            l2_normalize
  src/utils/Utils.py                   to_one_hot (+ create('softmax'))  -> f4_*.npz
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
             ('gru_mask', 5, 12, 300, 256, 150, True, 13), ('r18', 3, 49, 512, 128, 256, False, 14)]
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
    make_a1(probemb)
    make_a2(pie_model, tensor_utils)
    make_a34(losses_mod)
    make_a5()
    make_a6(eval_coco)
    import src.utils.Utils as utils_mod
    make_f4(losses_mod, utils_mod)
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
