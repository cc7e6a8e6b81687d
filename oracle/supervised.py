"""Oracle for SURVEY 8f-4: the client's supervised loss glue.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference (inline loop body of ClientTrainer.tra, not importable in isolation -- the module needs apex/torchvision):
  src/algorithms/ClientTrainer.py:344-351   one-hot margin, CE, centre loss, total
  src/algorithms/ClientTrainer.py:352-357   accuracy(fvec.data, labels, topk=(1, k))
  src/algorithms/ClientTrainer.py:114-129   accuracy()
  src/utils/Utils.py:6-13                   to_one_hot()
The criterion is nn.CrossEntropyLoss (src/losses/__init__.py:19, mean reduction).
Pinned by tests/golden/f4_*.npz (reference to_one_hot + criterion objects, literal statement sequence).
"""
import torch
import torch.nn.functional as F


def to_one_hot(y, n_dims):
    """Utils.py:6-13: zeros(B, n).scatter_(1, y.view(-1,1), 1)."""
    y = y.type(torch.LongTensor).view(-1, 1)
    return torch.zeros(y.size(0), n_dims).scatter_(1, y, 1)


def accuracy(output, target, topk=(1,)):
    """ClientTrainer.py:114-129: precision@k in percent, one 1-element tensor per k."""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / batch_size) for k in topk]


def supervised_glue(fvec, labels, class_weight, inter_distance, topk=5, center_weight=0.5):
    """ClientTrainer.py:344-357.  Returns (total, ce, center, prec1, preck); the precisions are computed on the
    margin-shifted logits, as the reference does (it rebinds `fvec` at :347 before calling accuracy)."""
    C = fvec.shape[1]
    one_hot = to_one_hot(labels, C).to(fvec.dtype)
    fvec = fvec - inter_distance * one_hot
    loss = F.cross_entropy(fvec, labels)
    center_labels = torch.arange(C, dtype=torch.long)
    center_loss = F.cross_entropy(torch.mm(class_weight, torch.t(class_weight)), center_labels)
    total = center_weight * center_loss + loss
    prec1, preck = accuracy(fvec.detach(), labels, topk=(1, topk))
    return total, loss, center_loss, prec1[0], preck[0]
