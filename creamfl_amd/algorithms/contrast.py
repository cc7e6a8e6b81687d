"""Host-side assembly of the client contrast step (rows A3 + A4) on top of the HIP ops.

Mirrors the loop bodies of the reference (not importable there: they are inline code):
  src/algorithms/ClientTrainer.py:369-429 (inter + intra), :431-480 (intra), :482-507 (inter)
  src/algorithms/MMClientTrainer.py:150-224, :225-293, :294-324
"""
from .. import ops

TEMPERATURE = 0.5        # hard-coded in the reference (ClientTrainer.py:388,411)


def client_contrast_loss(feature, global_same, global_other, d_idx, old_feature=None,
                         interintra_weight=0.5, loss_scale=False, use_inter=True, use_intra=True,
                         temperature=TEMPERATURE, root=False):
    """Uni-modal client.  Returns (loss, loss_inter | None, loss_moon | None).  root=True: the caller runs `loss.backward()` on
    the returned loss itself, as the reference does (ClientTrainer.py:420) -- see ops.client_contrast_fused.

    both  : (loss_moon + loss_inter) * w                                   ClientTrainer.py:417
            (loss_moon + loss_inter / (loss_inter/loss_moon).detach()) * w   :419 (--loss_scale)
    intra : loss_moon (:470)        inter : loss_inter (:502)
    """
    loss_inter = loss_moon = None
    if not (use_inter or use_intra):
        raise ValueError('no contrast term selected')
    bank = global_other if use_inter else global_same
    if ops.bank_attn_supported(feature.shape[0], bank.shape[0], feature.shape[1]):
        # one pass over the bank + one epilogue launch: both terms, their combination and all gradients (csrc/bank_attn.hip)
        loss, loss_inter, loss_moon, _, _ = ops.client_contrast_fused(
            feature, global_same, global_other, d_idx, old_feature, temperature, weight=interintra_weight,
            loss_scale=loss_scale, use_inter=use_inter, use_intra=use_intra, root=root)
        return loss, loss_inter, loss_moon
    if use_inter:
        loss_inter = ops.inter_contrast(feature, global_other, d_idx, temperature)[0]
    if use_intra:
        loss_moon = ops.intra_contrast(feature, global_same, d_idx, old_feature, temperature)
    if use_inter and use_intra:
        if not loss_scale:
            loss = (loss_moon + loss_inter) * interintra_weight
        else:
            loss = (loss_moon + loss_inter / (loss_inter / loss_moon).detach()) * interintra_weight
    elif use_intra:
        loss = loss_moon
    elif use_inter:
        loss = loss_inter
    else:
        raise ValueError('no contrast term selected')
    return loss, loss_inter, loss_moon


def mm_client_contrast_loss(out_img, out_txt, global_img, global_txt, d_idx, old_img=None, old_txt=None,
                            interintra_weight=0.5, loss_scale=False, use_inter=True, use_intra=True,
                            temperature=TEMPERATURE, root=False):
    """Multi-modal client (MMClientTrainer.py:150-324): the intra CE runs over the stacked [2B, 2]
    logits (mean over 2B rows), the inter term is CE(img vs G_txt) + CE(txt vs G_img)."""
    loss_inter = loss_intra = None
    if not (use_inter or use_intra):
        raise ValueError('no contrast term selected')
    b = out_img.shape[0]
    if ops.bank_attn_supported(b, global_img.shape[0], out_img.shape[1]):
        # per modality one pass over the bank + one finish launch (A4 inside); the second finish combines both modalities
        return ops.mm_client_contrast_fused(out_img, out_txt, global_img, global_txt, d_idx, old_img, old_txt, temperature,
                                            weight=interintra_weight, loss_scale=loss_scale, use_inter=use_inter,
                                            use_intra=use_intra, root=root)
    if use_intra:
        loss_intra = (ops.intra_contrast(out_img, global_img, d_idx, old_img, temperature, mean_divisor=2 * b)
                      + ops.intra_contrast(out_txt, global_txt, d_idx, old_txt, temperature, mean_divisor=2 * b))
    if use_inter:
        loss_inter = (ops.inter_contrast(out_img, global_txt, d_idx, temperature)[0]
                      + ops.inter_contrast(out_txt, global_img, d_idx, temperature)[0])
    if use_inter and use_intra:
        if not loss_scale:
            loss = (loss_intra + loss_inter) * interintra_weight
        else:
            loss = (loss_intra + loss_inter / (loss_inter / loss_intra).detach()) * interintra_weight
    elif use_intra:
        loss = loss_intra
    elif use_inter:
        loss = loss_inter
    else:
        raise ValueError('no contrast term selected')
    return loss, loss_inter, loss_intra
