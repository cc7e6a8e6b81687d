import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault('CFL_RUNTIME_TAG', 'tests')
import torch, numpy as np
import test_gpu_framework as T
from creamfl_amd.utils.synthetic import coco_batch
dev = torch.device('cuda:0')
b = coco_batch(int(os.environ.get('GB', '16')), dev, seed=21, bert=True)
batch = (b[0], b[1], None, b[3])
lf, gf, mf, state0 = T._bench_path_run(dev, False, 6, batch)
print('fused losses', lf)
def glob(g, h, pred=lambda n: True):
    a = torch.cat([g[n].double().flatten() for n in sorted(g) if pred(n)]); b = torch.cat([h[n].double().flatten() for n in sorted(g) if pred(n)])
    return float(torch.dot(a, b) / (a.norm() * b.norm())), float(a.norm() / b.norm())
def cmp(tag, g, l):
    print(tag, 'GLOBAL cos %.5f ratio %.4f | trunk cos %.5f | no-stem trunk cos %.5f | heads cos %.5f | text cos %.5f' % (glob(g, gf) + (glob(g, gf, lambda n: 'img_enc.cnn' in n)[0], glob(g, gf, lambda n: 'img_enc.cnn.layer' in n)[0], glob(g, gf, lambda n: 'cnn' not in n and 'txt_enc' not in n)[0], glob(g, gf, lambda n: 'txt_enc' in n)[0])))
    ag = T._grad_agreement(g, gf)
    print(tag, 'norm ratio range', min((g[n].norm() / (gf[n].norm() + 1e-30)).item() for n in g if ag[n][2] > 1e-3 * sum(v[2] ** 2 for v in ag.values()) ** 0.5), max((g[n].norm() / (gf[n].norm() + 1e-30)).item() for n in g if ag[n][2] > 1e-3 * sum(v[2] ** 2 for v in ag.values()) ** 0.5))
    total = sum(v[2] ** 2 for v in ag.values()) ** 0.5
    major = {n: v for n, v in ag.items() if v[2] > 1e-3 * total}
    worst = sorted(major.items(), key=lambda kv: kv[1][1])[:4]
    print(tag, 'WORST rel %.3g  WORST cos %.6f' % (max(v[0] for v in major.values()), min(v[1] for v in major.values())))
    print(tag, 'losses', [round(x, 3) for x in l], 'n_major', len(major), 'worst', [(n, round(v[0], 3), round(v[1], 4)) for n, v in worst])
    for n in ('linear.weight', 'img_enc.fc.weight', 'img_enc.pie_net.fc.weight', 'img_enc.cnn.layer4.2.conv3.weight', 'img_enc.cnn.layer4.2.bn3.weight', 'img_enc.cnn.layer4.2.bn3.bias', 'img_enc.cnn.layer4.0.conv1.weight', 'img_enc.cnn.layer3.0.bn1.bias','img_enc.cnn.layer1.0.bn1.bias', 'img_enc.cnn.conv1.weight'):
        if n in ag: print('   ', n, 'rel %.3g cos %.5f norm %.3g share %.3g' % (ag[n][0], ag[n][1], ag[n][2], ag[n][2]/total))
l2, g2, _, _ = T._bench_path_run(dev, False, 6, batch, state=state0)
cmp('fused again', g2, l2)
lu, gu, _, _ = T._bench_path_run(dev, True, 6, batch, state=state0)
cmp('unfused', gu, lu)
lu2, gu2, _, _ = T._bench_path_run(dev, True, 6, batch, state=state0)
ag = T._grad_agreement(gu2, gu); print('unfused vs unfused worst cos', min(v[1] for v in ag.values()))
l32, g32, _, _ = T._bench_path_run(dev, False, 1, batch, state=state0, fp32=True)
cmp('fp32', g32, l32)
ag = T._grad_agreement(gu, g32); total = sum(v[2] ** 2 for v in ag.values()) ** 0.5
print('unfused vs fp32 worst major cos', sorted([(round(v[1],4), n) for n, v in ag.items() if v[2] > 1e-3*total])[:5])
