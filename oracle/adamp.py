"""Oracle for the optimizer tail of row S1: AdamP restated from its paper.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call site: src/algorithms/optimizers.py:24 -> adamp.AdamP (adamp==0.3.0, requirements.txt), a
third-party package that is neither vendored in the reference nor installed here.  The algorithm below is
restated from Heo et al., "AdamP: Slowing Down the Slowdown for Momentum Optimizers on Scale-invariant
Weights" (ICLR 2021), Algorithm 2, with the package's defaults (delta = 0.1, wd_ratio = 0.1).
PARITY UNPINNED: there is no golden vector for it; the HIP implementation is checked against this file.
What IS known of the package beyond the paper and is restated here: its `step()` is a single loop over parameters, each
with its own state dict {'step', 'exp_avg', 'exp_avg_sq'}, so bias corrections follow every parameter's OWN step count
(parameters whose gradient is None are skipped and do not advance).
The projection test is evaluated with torch.where instead of a python `if` (same arithmetic).
"""
import math

import torch
import torch.nn.functional as F
from torch.optim.optimizer import Optimizer


class AdamP(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, delta=0.1, wd_ratio=0.1,
                 nesterov=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, delta=delta, wd_ratio=wd_ratio,
                        nesterov=nesterov)
        super().__init__(params, defaults)

    @staticmethod
    def _projection(p, grad, perturb, delta, wd_ratio, eps):
        """Channel-wise, then layer-wise scale-invariance test; returns (perturb, wd multiplier tensor)."""
        n0 = p.shape[0]
        expand = [-1] + [1] * (p.dim() - 1)
        pc, gc = p.reshape(n0, -1), grad.reshape(n0, -1)
        pl, gl = p.reshape(1, -1), grad.reshape(1, -1)
        cos_c = F.cosine_similarity(gc, pc, dim=1, eps=eps).abs().max()
        cos_l = F.cosine_similarity(gl, pl, dim=1, eps=eps).abs().max()
        hit_c = cos_c < delta / math.sqrt(pc.shape[1])
        hit_l = (~hit_c) & (cos_l < delta / math.sqrt(pl.shape[1]))
        # channel view
        pn_c = p / (pc.norm(dim=1).view(expand) + eps)
        proj_c = perturb - pn_c * (pn_c * perturb).reshape(n0, -1).sum(dim=1).view(expand)
        # layer view
        pn_l = p / (pl.norm() + eps)
        proj_l = perturb - pn_l * (pn_l * perturb).sum()
        out = torch.where(hit_c, proj_c, torch.where(hit_l, proj_l, perturb))
        wd = torch.where(hit_c | hit_l, torch.full_like(cos_c, wd_ratio), torch.ones_like(cos_c))
        return out, wd

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            params, grads, avgs, sqs = [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                state['step'] += 1
                params.append(p); grads.append(p.grad); avgs.append(state['exp_avg']); sqs.append(state['exp_avg_sq'])
            if not params:
                continue
            # `state['step']` is kept PER PARAMETER (as adamp==0.3.0 does: its step() is one loop over parameters, each with
            # its own state dict): a parameter whose gradient was None in earlier steps -- the criterion's shift /
            # negative_scale during every KD phase (MMFL.py:385-391 back-propagates an MSE that does not reach them), a whole
            # tower when only one kind of client exists -- lags behind, and its bias corrections use ITS count.
            steps = [self.state[p]['step'] for p in params]
            torch._foreach_mul_(avgs, beta1)
            torch._foreach_add_(avgs, grads, alpha=1 - beta1)
            torch._foreach_mul_(sqs, beta2)
            torch._foreach_addcmul_(sqs, grads, grads, value=1 - beta2)
            denoms = list(torch._foreach_sqrt(sqs))
            for i, st in enumerate(steps):
                denoms[i].div_(math.sqrt(1 - beta2 ** st)).add_(group['eps'])
            if group['nesterov']:
                perturbs = torch._foreach_mul(avgs, beta1)
                torch._foreach_add_(perturbs, grads, alpha=1 - beta1)
                torch._foreach_div_(perturbs, denoms)
            else:
                perturbs = torch._foreach_div(avgs, denoms)
            perturbs = list(perturbs)
            for i, p in enumerate(params):
                wd_mul = None
                if p.dim() > 1:
                    perturbs[i], wd_mul = self._projection(p, grads[i], perturbs[i], group['delta'], group['wd_ratio'],
                                                           group['eps'])
                if group['weight_decay'] > 0:
                    p.mul_(1 - group['lr'] * group['weight_decay'] * (1 if wd_mul is None else wd_mul))
                p.add_(perturbs[i], alpha=-group['lr'] / (1 - beta1 ** steps[i]))
        return loss
