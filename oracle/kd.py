"""Oracle for row (f1): the KD distillation terms of the server.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: src/algorithms/MMFL.py:296 (client_loss_cri = nn.MSELoss()), :355-378 (code_sim and the three
`if self.args.num_*_clients > 0` blocks).  The image term is added once per image-client block AND once in the
multimodal block, i.e. twice when both kinds of client exist -- reproduced, not fixed.
Pinned by tests/golden/kd_*.npz (literal statement sequence evaluated in tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F


def code_sim(output, target):
    """MMFL.py:355-359."""
    output = output.sum(dim=1) if output.dim() == 3 else output
    return F.mse_loss(output, target.type_as(output))


def kd_loss(out_img, out_txt, img_vec, txt_vec, d_idx, num_img_clients, num_txt_clients, num_mm_clients, kd_weight):
    """MMFL.py:361-378.  d_idx: positions of this batch inside the aggregated [M, D] representations."""
    d_idx = torch.as_tensor(d_idx, dtype=torch.long)
    loss = 0
    if num_img_clients > 0:
        loss = loss + kd_weight * code_sim(out_img, img_vec[d_idx, :])
    if num_txt_clients > 0:
        loss = loss + kd_weight * code_sim(out_txt, txt_vec[d_idx, :])
    if num_mm_clients > 0:
        loss = loss + kd_weight * code_sim(out_img, img_vec[d_idx, :])
        loss = loss + kd_weight * code_sim(out_txt, txt_vec[d_idx, :])
    return loss
