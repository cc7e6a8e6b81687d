"""torch.autograd wrappers over the C ABI (include/creamfl_hip.h).

Every function here runs on the current HIP stream of the tensors' device, takes and returns
ordinary torch tensors, and raises if the HIP library is missing or an argument is not a
contiguous fp32 device tensor -- there is no CPU fallback (the CPU oracle lives in oracle/ and is
test infrastructure only).
"""
import contextlib
import threading as _threading
import ctypes
from collections.abc import Mapping

import torch

from . import _lib


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _raw_stream(device):
    """The current HIP stream of `device` as an integer handle.  torch.cuda.current_stream() builds a Stream object per call
    (~6 us of host time, paid by every one of the ~650 hand-written launches of a server step); the raw getter is ~0.3 us."""
    if _RAW_STREAM is not None:
        idx = device.index
        return _RAW_STREAM(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t):
    return _raw_stream(t.device)


def _ptr(t):
    """Device address as a plain int (None = NULL): every entry point has c_void_p argtypes (_lib.SIGNATURES, checked against the
    header by tests/test_abi.py), so ctypes converts; building a c_void_p object per argument was ~3 500 objects per server step."""
    return t.data_ptr() if t is not None else None


def _f32(t, name):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.CreamflHipError(f'{name}: expected a CUDA/HIP tensor (no CPU fallback in creamfl_amd)')
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------- A1: pair loss
class LazyLossDict(Mapping):
    """The reference returns 11 python floats (11 device syncs, probemb.py:245-255).  This mapping
    has the same keys and values but copies the 8-float statistics block to the host only when a
    value is first read."""
    KEYS = ('i2t_loss', 't2i_loss', 'i2t_pos_loss', 'i2t_neg_loss', 't2i_pos_loss', 't2i_neg_loss',
            'uniform_loss', 'vib_loss', 'shift', 'negative_scale', 'loss')

    def __init__(self, out8, shift, negative_scale):
        self._dev = (out8, shift, negative_scale)
        self._host = None

    def _materialise(self):
        if self._host is None:
            out8, shift, ns = self._dev
            v = torch.cat([out8.detach(), shift.detach().reshape(1), ns.detach().reshape(1)]).cpu().tolist()
            loss, pos, neg = v[0], v[1], v[2]
            self._host = {'i2t_loss': pos + neg, 't2i_loss': pos + neg, 'i2t_pos_loss': pos, 'i2t_neg_loss': neg,
                          't2i_pos_loss': pos, 't2i_neg_loss': neg, 'uniform_loss': 0, 'vib_loss': 0,
                          'shift': v[8], 'negative_scale': v[9], 'loss': loss}
        return self._host

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)


class _PairLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, I, T, a, b, eps):
        lib = _lib.load()
        N, D = I.shape
        need_grad = any(ctx.needs_input_grad[:4])
        out8 = torch.empty(8, dtype=torch.float32, device=I.device)
        coef = torch.empty(2, N, N, dtype=torch.float32, device=I.device) if need_grad else None
        ws = _ws(lib.cfl_pair_loss_ws_bytes(N, D), I.device)
        _lib.check(lib.cfl_pair_loss_fwd(_ptr(I), _ptr(T), N, D, _ptr(a), _ptr(b), eps, _ptr(out8), _ptr(coef),
                                         _ptr(ws), _stream(I)), 'cfl_pair_loss_fwd')
        ctx.save_for_backward(I, T, coef if coef is not None else out8, out8, ws)
        ctx.has_coef = need_grad
        ctx.mark_non_differentiable(out8)
        return out8[0].clone(), out8

    @staticmethod
    def backward(ctx, gloss, _gstats):
        lib = _lib.load()
        I, T, coef, out8, ws = ctx.saved_tensors
        if not ctx.has_coef:
            raise _lib.CreamflHipError('pair_loss backward without saved coefficients')
        N, D = I.shape
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dI = torch.empty_like(I)
        dT = torch.empty_like(T)
        _lib.check(lib.cfl_pair_loss_bwd(_ptr(I), _ptr(T), _ptr(coef), N, D, _ptr(g), _ptr(dI), _ptr(dT), _ptr(ws),
                                         _stream(I)), 'cfl_pair_loss_bwd')
        da = (out8[3] * g[0]).reshape(1)
        db = (out8[4] * g[0]).reshape(1)
        return dI, dT, da, db, None


def pair_loss(image_features, caption_features, negative_scale, shift, eps=1e-6):
    """A1 (src/criterions/probemb.py:221-256).  Returns (loss 0-d tensor with grad, stats[8] tensor =
    {loss, pos, neg, dL/da, dL/db, 0, 0, 0})."""
    I = _f32(image_features, 'image_features')
    T = _f32(caption_features, 'caption_features')
    if I.dim() != 2 or I.shape != T.shape:
        raise RuntimeError('# anchors ({}) != # candidates ({})'.format(tuple(I.shape), tuple(T.shape)))
    a = _f32(negative_scale, 'negative_scale').reshape(1)
    b = _f32(shift, 'shift').reshape(1)
    return _PairLossFn.apply(I, T, a, b, float(eps))


# --------------------------------------------------------------------------- A3/A4: client contrast
# Persistent per-device state of the fused path: cached workspaces (keyed by size) and the zero-initialised election
# counter of the epilogue kernel (include/creamfl_hip.h: `sync` must be 0 before the first call and is left 0).
_CONTRAST_STATE = {}


def _contrast_state(device, ws_bytes):
    st = _CONTRAST_STATE.get(device)
    if st is None:
        st = _CONTRAST_STATE[device] = {'sync': torch.zeros(1, dtype=torch.int32, device=device), 'ws': None, 'n': 0}
    if st['n'] < ws_bytes:
        st['ws'] = torch.empty(max(int(ws_bytes), 256), dtype=torch.uint8, device=device)
        st['n'] = st['ws'].numel()
    return st


import os as _os
_BANK_EXACT = bool(_os.environ.get('CFL_BANK_EXACT'))       # A/B switch: the exact-fp32 two-pass kernels (csrc/bank.hip), the one reference path

# Pre-split bank images (csrc/bank_gsplit.h).  The global banks are frozen while a client trains (ClientTrainer.py:369-372:
# one pair of global feature tensors per round, hundreds of steps against them), so the fp32 -> (bf16 hi, bf16 lo) image is
# built once per bank VERSION: an entry is reused while the same storage (kept alive by the entry, so its address cannot be
# handed to another tensor) still carries the version counter it was built from.
_BANK_IMAGES = {}            # (data_ptr, M, D) -> [source tensor, version, image, last-use tick]
_BANK_IMAGES_MAX = 4         # the two live banks of a round (image + text) and, briefly, the next round's two
_bank_tick = [0]
_CONW_NOIMG = bool(_os.environ.get('CFL_CONW_NOIMG'))     # A/B: the tile GEMM of bank.hip for every con_w shape
BANK_IMAGE_BUILDS = [0]      # how many images were built (tests / benches read it)


def invalidate_bank_images():
    """Drop every cached bank image.  MMFL calls this where it replaces global_img_feature / global_txt_feature each round
    (MMFL.py:194-221), so that the previous round's banks and their images (2 x the bank's bytes each) are released at once.
    Also the remedy after a write the version counter cannot see (`.data`, raw-pointer kernels): see bank_image()."""
    _BANK_IMAGES.clear()


def bank_image(G):
    """The pre-split image of a [M, D] fp32 bank (device tensor of cfl_bank_image_bytes bytes), cached per bank version.
    An entry keeps its source tensor alive (so the address in the key cannot be handed to another tensor) -- at most
    _BANK_IMAGES_MAX entries, and MMFL drops them all when it replaces the global features (invalidate_bank_images).
    Validity is the tensor's autograd version counter: in-place torch operations bump it; writes through `.data` or through
    raw-pointer kernels do not -- call invalidate_bank_images() after those."""
    lib = _lib.load()
    M, D = G.shape
    key = (G.data_ptr(), M, D)
    _bank_tick[0] += 1
    ent = _BANK_IMAGES.get(key)
    if ent is not None and ent[1] == G._version:
        ent[3] = _bank_tick[0]
        return ent[2]
    img = ent[2] if ent is not None else torch.empty(int(lib.cfl_bank_image_bytes(M, D)), dtype=torch.uint8, device=G.device)
    _lib.check(lib.cfl_bank_image_build(G.data_ptr(), M, D, img.data_ptr(), _raw_stream(G.device)),
               'cfl_bank_image_build')
    BANK_IMAGE_BUILDS[0] += 1
    _BANK_IMAGES[key] = [G, G._version, img, _bank_tick[0]]
    if len(_BANK_IMAGES) > _BANK_IMAGES_MAX:
        del _BANK_IMAGES[min(_BANK_IMAGES, key=lambda k: _BANK_IMAGES[k][3])]
    return img


def bank_image_supported(B, M, D):
    return (not _BANK_EXACT) and _bank_plan(int(B), int(M), int(D), False) is not None


_BANK_PLAN = {}          # (B, M, D, need_grad) -> workspace bytes, or None when the fused path does not take the shape


def _bank_plan(B, M, D, need):
    key = (B, M, D, need)
    v = _BANK_PLAN.get(key, -1)
    if v == -1:
        lib = _lib.load()
        v = int(lib.cfl_bank_gsplit_ws_bytes(B, M, D, int(need))) if lib.cfl_bank_gsplit_supported(B, M, D) else None
        _BANK_PLAN[key] = v
    return v


def bank_attn_supported(B, M, D):
    """True when the single-pass 3 x bf16-split kernels take this shape: D <= 768, D % 4 == 0, on a pre-split bank image
    (csrc/bank_gsplit.h).  Other widths run the exact-fp32 two-pass kernels (csrc/bank.hip)."""
    return bank_image_supported(B, M, D)


class _ClientContrastFn(torch.autograd.Function):
    """Fused A3 (+ A4): one pass over the bank gives the log-sum-exp and the unit gradient; one finish launch gives the
    positive dots, the intra term, the means and the combined loss (csrc/bank_attn.hip).  Host work per call: three
    allocations and one ctypes call (the workspace, its size and the election counter are cached per device / shape)."""

    @staticmethod
    def forward(ctx, F, G_other, G_same, idx, F_old, inv_tau, weight, mode, b_div, root=False):
        lib = _lib.load()
        B, D = F.shape
        M = (G_other if (mode & 1) else G_same).shape[0]
        dev = F.device
        need = ctx.needs_input_grad[0]
        # root (the caller backpropagates from THIS loss: upstream gradient 1) and no --loss_scale: the finish launch writes the final
        # gradient and the backward pass launches nothing (include/creamfl_hip.h, want_grad = 2)
        direct = bool(need and root and not (mode & 4))
        out = torch.empty(8, dtype=torch.float32, device=dev)          # out5 = out[0:5]; the differentiable loss = out[5]
        aux = torch.empty(2, B, dtype=torch.float32, device=dev) if (mode & 1) else None        # lse, pos
        dFs = None
        if need:                                                       # unit gradients (inter, moon) | the final gradient
            dFs = torch.empty((B, D) if direct else (2, B, D), dtype=torch.float32, device=dev)
        st = _contrast_state(dev, _bank_plan(B, M, D, bool(need)))
        p_out = out.data_ptr()
        p_aux = aux.data_ptr() if aux is not None else 0
        p_dfs = dFs.data_ptr() if need else 0
        tail = (B, M, D, b_div, inv_tau, weight, mode, 2 if direct else int(need),
                p_out, p_aux, p_aux + 4 * B if p_aux else 0, p_dfs if (need and ((mode & 1) or direct)) else 0,
                p_dfs + 4 * B * D if (need and (mode & 2) and not direct) else 0, st['ws'].data_ptr(), st['sync'].data_ptr(),
                _raw_stream(dev))
        p_same = G_same.data_ptr() if G_same is not None else 0
        p_old = F_old.data_ptr() if F_old is not None else 0
        # bank pass on the pre-split image of G_other (built once per bank version), 32-row groups (csrc/bank_gsplit.h); with the
        # inter term off only the finish launch runs
        _lib.check(lib.cfl_client_contrast_img_fwd(F.data_ptr(), bank_image(G_other).data_ptr() if (mode & 1) else 0,
                                                   G_other.data_ptr() if G_other is not None else 0, p_same,
                                                   idx.data_ptr(), p_old, *tail), 'cfl_client_contrast_img_fwd')
        ctx.mode = mode
        ctx.save_for_backward(out, dFs if need else out)
        ctx.has = need
        ctx.direct = direct
        ctx.mark_non_differentiable(out)
        if aux is not None:
            ctx.mark_non_differentiable(aux)
        return out.new_empty(()).set_(out.untyped_storage(), 5, ()), out, aux

    @staticmethod
    def backward(ctx, gloss, _g5, _gaux):
        lib = _lib.load()
        out, dFs = ctx.saved_tensors
        if not ctx.has:
            raise _lib.CreamflHipError('client_contrast backward without saved gradients')
        if ctx.direct:                         # the caller declared the loss the root: gloss is the 1 of loss.backward()
            return dFs, None, None, None, None, None, None, None, None, None
        _, B, D = dFs.shape
        g = gloss if (gloss.dtype == torch.float32 and gloss.is_contiguous()) else gloss.to(torch.float32).contiguous()
        dF = torch.empty(B, D, dtype=torch.float32, device=dFs.device)
        p = dFs.data_ptr()
        _lib.check(lib.cfl_client_contrast_bwd(p if (ctx.mode & 1) else 0, p + 4 * B * D if (ctx.mode & 2) else 0, out.data_ptr(),
                                               g.data_ptr(), B, D, dF.data_ptr(), _raw_stream(dF.device)),
                   'cfl_client_contrast_bwd')
        return dF, None, None, None, None, None, None, None, None, None


def client_contrast_fused(feature, global_same, global_other, d_idx, old_feature, temperature=0.5, weight=1.0, loss_scale=False,
                          use_inter=True, use_intra=True, mean_divisor=None, root=False):
    """Rows A3 + A4 and their combination (ClientTrainer.py:386-419) in two launches (bank pass, finish).
    Returns (loss, loss_inter | None, loss_moon | None, lse | None, pos | None).  Needs bank_attn_supported(B, M, D).
    root=True is the caller's promise that it calls `loss.backward()` on the returned loss itself (ClientTrainer.py:420): the upstream
    gradient is then the constant 1, the finish launch writes the final feature gradient and the backward launches nothing.  A loss
    that is scaled or summed into something else before the backward pass must keep root=False (--loss_scale always does)."""
    F = _f32(feature, 'feature')
    mode = (1 if use_inter else 0) | (2 if use_intra else 0) | (4 if loss_scale else 0)
    if not (mode & 3):
        raise ValueError('no contrast term selected')
    Go = _f32(global_other.detach(), 'global_other') if use_inter else None
    Gs = _f32(global_same.detach(), 'global_same') if use_intra else None
    Fo = _f32(old_feature.detach(), 'old_feature') if use_intra else None
    for G in (Go, Gs):
        if G is not None and (G.dim() != 2 or F.dim() != 2 or G.shape[1] != F.shape[1]):
            raise RuntimeError(f'shape mismatch {tuple(F.shape)} vs {tuple(G.shape)}')
    if Go is not None and Gs is not None and Go.shape != Gs.shape:
        raise RuntimeError(f'the two global banks differ in shape: {tuple(Go.shape)} vs {tuple(Gs.shape)}')
    if Fo is not None and Fo.shape != F.shape:
        raise RuntimeError(f'shape mismatch {tuple(F.shape)} vs {tuple(Fo.shape)}')
    loss, out, aux = _ClientContrastFn.apply(F, Go, Gs, _idx(d_idx, F.device), Fo, 1.0 / float(temperature), float(weight), mode,
                                            int(mean_divisor) if mean_divisor else F.shape[0], bool(root))
    return (loss, out[1] if use_inter else None, out[2] if use_intra else None,
            aux[0] if aux is not None else None, aux[1] if aux is not None else None)


class _MMClientContrastFn(torch.autograd.Function):
    """The multi-modal client's contrast step (MMClientTrainer.py:164-206, :246-264, :301-308) as two chained calls of the
    fused uni-modal step: image rows against (G_txt bank, G_img positives), then caption rows against (G_img bank, G_txt
    positives) with mode bit 3, whose finish launch adds the first call's terms before combining -- 2 bank passes + 2 finish
    launches forward, ONE launch backward for both modalities (they share the coefficient pair in `out`)."""

    @staticmethod
    def forward(ctx, F_img, F_txt, G_img, G_txt, idx, Fo_img, Fo_txt, inv_tau, weight, mode, root=False):
        lib = _lib.load()
        B, D = F_img.shape
        M = G_img.shape[0]
        dev = F_img.device
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        direct = bool(need and root and not (mode & 4))       # see _ClientContrastFn: the final gradients from the finish launches
        out = torch.empty(8, dtype=torch.float32, device=dev)
        aux = torch.empty(2, 2, B, dtype=torch.float32, device=dev) if (mode & 1) else None     # [modality][lse, pos][B]
        dFs = None
        if need:                                              # [inter, moon][modality][B, D] | direct: [modality][B, D]
            dFs = torch.empty((2, B, D) if direct else (2, 2, B, D), dtype=torch.float32, device=dev)
        st = _contrast_state(dev, _bank_plan(B, M, D, bool(need)))
        stream = _raw_stream(dev)
        for k, (F, G_other, G_same, F_old) in enumerate(((F_img, G_txt, G_img, Fo_img), (F_txt, G_img, G_txt, Fo_txt))):
            p_aux = aux.data_ptr() + 8 * B * k if aux is not None else 0
            p_dfi = dFs.data_ptr() + 4 * B * D * k if (need and ((mode & 1) or direct)) else 0
            p_dfm = dFs.data_ptr() + 4 * B * D * (2 + k) if (need and (mode & 2) and not direct) else 0
            tail = (B, M, D, 2 * B, inv_tau, weight, mode | (8 if k else 0), 2 if direct else int(need), out.data_ptr(), p_aux,
                    p_aux + 4 * B if p_aux else 0, p_dfi, p_dfm, st['ws'].data_ptr(), st['sync'].data_ptr(), stream)
            p_other = G_other.data_ptr() if (mode & 1) else 0
            p_same = G_same.data_ptr() if (mode & 2) else 0
            p_old = F_old.data_ptr() if (mode & 2) else 0
            _lib.check(lib.cfl_client_contrast_img_fwd(F.data_ptr(), bank_image(G_other).data_ptr() if (mode & 1) else 0,
                                                       p_other, p_same, idx.data_ptr(), p_old, *tail),
                       'cfl_client_contrast_img_fwd')
        ctx.mode = mode
        ctx.save_for_backward(out, dFs if need else out)
        ctx.has = need
        ctx.direct = direct
        ctx.mark_non_differentiable(out)
        if aux is not None:
            ctx.mark_non_differentiable(aux)
        return out.new_empty(()).set_(out.untyped_storage(), 5, ()), out, aux

    @staticmethod
    def backward(ctx, gloss, _g5, _gaux):
        lib = _lib.load()
        out, dFs = ctx.saved_tensors
        if not ctx.has:
            raise _lib.CreamflHipError('mm_client_contrast backward without saved gradients')
        if ctx.direct:
            return dFs[0], dFs[1], None, None, None, None, None, None, None, None, None
        _, _, B, D = dFs.shape
        g = gloss if (gloss.dtype == torch.float32 and gloss.is_contiguous()) else gloss.to(torch.float32).contiguous()
        dF = torch.empty(2, B, D, dtype=torch.float32, device=dFs.device)
        p = dFs.data_ptr()
        _lib.check(lib.cfl_client_contrast_bwd(p if (ctx.mode & 1) else 0, p + 8 * B * D if (ctx.mode & 2) else 0, out.data_ptr(),
                                               g.data_ptr(), 2 * B, D, dF.data_ptr(), _raw_stream(dF.device)),
                   'cfl_client_contrast_bwd')
        return dF[0], dF[1], None, None, None, None, None, None, None, None, None


def mm_client_contrast_fused(out_img, out_txt, global_img, global_txt, d_idx, old_img=None, old_txt=None, temperature=0.5,
                             weight=1.0, loss_scale=False, use_inter=True, use_intra=True, root=False):
    """The multi-modal client's contrast terms and their combination (MMClientTrainer.py:164-206; intra only :246-264; inter only
    :301-308): 4 launches forward, 1 backward.  Returns (loss, loss_inter | None, loss_intra | None).  Needs
    bank_attn_supported(B, M, D)."""
    Fi, Ft = _f32(out_img, 'out_img'), _f32(out_txt, 'out_txt')
    mode = (1 if use_inter else 0) | (2 if use_intra else 0) | (4 if loss_scale else 0)
    if not (mode & 3):
        raise ValueError('no contrast term selected')
    Gi, Gt = _f32(global_img.detach(), 'global_img'), _f32(global_txt.detach(), 'global_txt')
    if Fi.dim() != 2 or Fi.shape != Ft.shape or Gi.dim() != 2 or Gi.shape != Gt.shape or Gi.shape[1] != Fi.shape[1]:
        raise RuntimeError(f'shape mismatch {tuple(Fi.shape)} / {tuple(Ft.shape)} vs {tuple(Gi.shape)} / {tuple(Gt.shape)}')
    Oi = Ot = None
    if use_intra:
        Oi, Ot = _f32(old_img.detach(), 'old_img'), _f32(old_txt.detach(), 'old_txt')
        if Oi.shape != Fi.shape or Ot.shape != Ft.shape:
            raise RuntimeError(f'shape mismatch {tuple(Fi.shape)} vs {tuple(Oi.shape)} / {tuple(Ot.shape)}')
    loss, out, _ = _MMClientContrastFn.apply(Fi, Ft, Gi, Gt, _idx(d_idx, Fi.device), Oi, Ot, 1.0 / float(temperature), float(weight),
                                            mode, bool(root))
    return loss, out[1] if use_inter else None, out[2] if use_intra else None


class _BankInterFn(torch.autograd.Function):
    """Exact-fp32 two-pass path (v_mfma_f32_32x32x2_f32): D > 768 / D % 4 != 0, and the one A/B reference (CFL_BANK_EXACT=1)."""

    @staticmethod
    def forward(ctx, F, G, idx, inv_tau):
        lib = _lib.load()
        B, D = F.shape
        M = G.shape[0]
        dev = F.device
        lse = torch.empty(B, dtype=torch.float32, device=dev)
        pos = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        logits_t = torch.empty(M, B, dtype=torch.float32, device=dev) if need else None
        ws = _ws(lib.cfl_bank_ws_bytes(B, M, D), dev)
        _lib.check(lib.cfl_bank_lse_fwd(_ptr(F), _ptr(G), _ptr(idx), B, M, D, inv_tau, _ptr(lse), _ptr(pos), _ptr(loss),
                                        _ptr(logits_t), _ptr(ws), _stream(F)), 'cfl_bank_lse_fwd')
        ctx.save_for_backward(G, idx, lse, logits_t if need else lse, ws)
        ctx.inv_tau = inv_tau
        ctx.shape = (B, M, D)
        ctx.mark_non_differentiable(lse, pos)
        return loss[0].clone(), lse, pos

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        lib = _lib.load()
        G, idx, lse, logits_t, ws = ctx.saved_tensors
        B, M, D = ctx.shape
        g = gloss.reshape(1).to(torch.float32).contiguous()
        dF = torch.empty(B, D, dtype=torch.float32, device=G.device)
        _lib.check(lib.cfl_bank_lse_bwd(_ptr(logits_t), _ptr(G), _ptr(idx), _ptr(lse), B, M, D, ctx.inv_tau, _ptr(g),
                                        _ptr(dF), _ptr(ws), _stream(G)), 'cfl_bank_lse_bwd')
        return dF, None, None, None


def _idx(d_idx, device):
    """int64 device tensor of bank positions.  A device tensor passes through without any host work; host sequences are
    range-checked here (the kernels treat an out-of-range index as 'no positive', never as an address)."""
    if torch.is_tensor(d_idx):
        return d_idx.to(device=device, dtype=torch.int64).contiguous()
    return torch.as_tensor(list(d_idx) if not isinstance(d_idx, (list, tuple)) else d_idx,
                           dtype=torch.int64, device=device)


def inter_contrast(feature, global_other, d_idx, temperature=0.5):
    """A3: CrossEntropy(feature @ global_other.T / temperature, d_idx)  (ClientTrainer.py:388,400-401).
    Returns (loss, lse[B], pos[B])."""
    F = _f32(feature, 'feature')
    G = _f32(global_other.detach(), 'global_other')
    if F.dim() != 2 or G.dim() != 2 or F.shape[1] != G.shape[1]:
        raise RuntimeError(f'shape mismatch {tuple(F.shape)} vs {tuple(G.shape)}')
    if bank_attn_supported(F.shape[0], G.shape[0], F.shape[1]):
        loss, _, _, lse, pos = client_contrast_fused(F, None, G, d_idx, None, temperature, use_inter=True, use_intra=False)
        return loss, lse, pos
    return _BankInterFn.apply(F, G, _idx(d_idx, F.device), 1.0 / float(temperature))


class _IntraFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, Gs, idx, Fo, inv_tau, b_div):
        lib = _lib.load()
        B, D = F.shape
        loss = torch.empty(1, dtype=torch.float32, device=F.device)
        need = ctx.needs_input_grad[0]
        dF = torch.empty_like(F) if need else None
        ws = _ws(lib.cfl_intra_ws_bytes(B), F.device)
        _lib.check(lib.cfl_intra_fwd(_ptr(F), _ptr(Gs), _ptr(idx), _ptr(Fo), B, D, Gs.shape[0], b_div, inv_tau, _ptr(loss), _ptr(dF),
                                     _ptr(ws), _stream(F)), 'cfl_intra_fwd')
        ctx.save_for_backward(dF if need else loss)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        (dF,) = ctx.saved_tensors
        return dF * gloss, None, None, None, None, None


def intra_contrast(feature, global_same, d_idx, old_feature, temperature=0.5, mean_divisor=None):
    """A4: CE([<f, G_same[idx]>, <f, f_old>] / temperature, 0)  (ClientTrainer.py:404-414).
    `mean_divisor` overrides the CE mean divisor (2B when two modalities are stacked,
    MMClientTrainer.py:184-188)."""
    F = _f32(feature, 'feature')
    Gs = _f32(global_same.detach(), 'global_same')
    Fo = _f32(old_feature.detach(), 'old_feature')
    if F.shape != Fo.shape:
        raise RuntimeError(f'shape mismatch {tuple(F.shape)} vs {tuple(Fo.shape)}')
    return _IntraFn.apply(F, Gs, _idx(d_idx, F.device), Fo, 1.0 / float(temperature),
                          int(mean_divisor) if mean_divisor else F.shape[0])


class _KdMseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, agg, idx, weight):
        lib = _lib.load()
        B, D = out.shape
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        need = ctx.needs_input_grad[0]
        dout = torch.empty_like(out) if need else None
        ws = _ws(lib.cfl_intra_ws_bytes(B), out.device)
        _lib.check(lib.cfl_kd_mse(_ptr(out), _ptr(agg), _ptr(idx), B, D, agg.shape[0], weight, _ptr(loss), _ptr(dout),
                                  _ptr(ws), _stream(out)), 'cfl_kd_mse')
        ctx.save_for_backward(dout if need else loss)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None, None


def kd_mse(output, aggregate, d_idx, kd_weight=1.0):
    """kd_weight * MSELoss(output, aggregate[d_idx]) without materialising the gathered target (MMFL.py:352-378)."""
    out = _f32(output, 'output')
    agg = _f32(aggregate.detach(), 'aggregate')
    if out.dim() != 2 or agg.dim() != 2 or out.shape[1] != agg.shape[1]:
        raise RuntimeError(f'shape mismatch {tuple(out.shape)} vs {tuple(agg.shape)}')
    return _KdMseFn.apply(out, agg, _idx(d_idx, out.device), float(kd_weight))


# ---- A2c: the text towers' GRU, last valid step only (gru.hip) ---------------------------------------------------------------
# The reference keeps ONE vector per caption of its packed bidirectional GRU: gather(output, lengths - 1) = [the forward
# direction's final state | the backward direction's FIRST step (a cell on the last word from a zero state)]
# (language_model.py:93-107, caption_encoder.py:87-101).  The recurrence is one launch per autograd direction, the lengths stay
# on the device, and the four weight / input gradients are library GEMMs over the pre-activation gradients the kernel writes.
GRU_FUSED = [_os.environ.get('CFL_NO_GRU_FUSED', '0') != '1']


class _GruLastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, words, lens, w_ih, w_hh, b_ih, b_hh, track=True):
        lib = _lib.load()
        B, T, E = words.shape
        H = w_hh.shape[1]
        x2 = words.reshape(B * T, E)
        xp = torch.addmm(b_ih, x2, w_ih.t())                               # [B * T, 3H]: the input side of every step at once
        # (needs_input_grad is True under no_grad() too whenever a weight has requires_grad, and grad mode is always OFF inside a
        # Function's forward: the caller passes its own grad mode as `track` -- the frozen old model's forward of the intra contrast
        # must not write and keep the [T+1, B, H] states and [B, T, 4H] gates)
        need = bool(track) and any(ctx.needs_input_grad)
        out = torch.empty(B, H, dtype=torch.float32, device=words.device)
        hs = torch.empty(T + 1, B, H, dtype=torch.float32, device=words.device) if need else None
        gates = torch.empty(B, T, 4 * H, dtype=torch.float32, device=words.device) if need else None
        # widths beyond the register-resident kernels stream W_hh from L2 every step: k-major copy for coalesced column reads
        w_hh_t = w_hh.t().contiguous() if lib.cfl_gru_streams_weights(H) else None
        _lib.check(lib.cfl_gru_fwd(_ptr(xp), _ptr(w_hh), _ptr(w_hh_t), _ptr(b_hh), _ptr(lens), _ptr(out), _ptr(hs), _ptr(gates),
                                   B, T, H, _stream(words)), 'cfl_gru_fwd')
        if need:
            ctx.save_for_backward(x2, lens, w_ih, w_hh, hs, gates)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x2, lens, w_ih, w_hh, hs, gates = ctx.saved_tensors
        T, B, H = hs.shape[0] - 1, hs.shape[1], hs.shape[2]
        dxp = torch.empty(B * T, 3 * H, dtype=torch.float32, device=x2.device)
        dg = torch.empty(T * B, 3 * H, dtype=torch.float32, device=x2.device)
        _lib.check(lib.cfl_gru_bwd(_ptr(dout.contiguous()), _ptr(w_hh), _ptr(lens), _ptr(hs), _ptr(gates), _ptr(dxp), _ptr(dg),
                                   B, T, H, _stream(x2)), 'cfl_gru_bwd')
        need = ctx.needs_input_grad
        dwords = (dxp @ w_ih).view(B, T, -1) if need[0] else None
        dw_ih = dxp.t() @ x2 if need[2] else None
        dw_hh = dg.t() @ hs[:T].view(T * B, H) if need[3] else None
        db_ih = dxp.sum(0) if need[4] else None
        db_hh = dg.sum(0) if need[5] else None
        return dwords, None, dw_ih, dw_hh, db_ih, db_hh, None


class _GruCell0Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_last, w_ih, w_hh, b_ih, b_hh, track=True):
        lib = _lib.load()
        B, H = x_last.shape[0], w_hh.shape[1]
        gx = torch.addmm(b_ih, x_last, w_ih.t())
        need = bool(track) and any(ctx.needs_input_grad)
        out = torch.empty(B, H, dtype=torch.float32, device=x_last.device)
        saved = torch.empty(B, 3 * H, dtype=torch.float32, device=x_last.device) if need else None
        _lib.check(lib.cfl_gru_cell0_fwd(_ptr(gx), _ptr(b_hh), _ptr(out), _ptr(saved), B, H, _stream(x_last)), 'cfl_gru_cell0_fwd')
        if need:
            ctx.save_for_backward(x_last, w_ih, w_hh, b_hh, saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x_last, w_ih, w_hh, b_hh, saved = ctx.saved_tensors
        B, H = x_last.shape[0], w_hh.shape[1]
        dgx = torch.empty(B, 3 * H, dtype=torch.float32, device=x_last.device)
        dgh = torch.empty(B, 3 * H, dtype=torch.float32, device=x_last.device)
        _lib.check(lib.cfl_gru_cell0_bwd(_ptr(dout.contiguous()), _ptr(saved), _ptr(b_hh), _ptr(dgx), _ptr(dgh), B, H,
                                         _stream(x_last)), 'cfl_gru_cell0_bwd')
        need = ctx.needs_input_grad
        dx = dgx @ w_ih if need[0] else None
        dw_ih = dgx.t() @ x_last if need[1] else None
        # the state this cell's W_hh multiplies is zero: its gradient is a zero TENSOR, not None (the reference's packed GRU returns
        # zeros too, and an optimizer with weight decay treats the two differently)
        dw_hh = torch.zeros_like(w_hh) if need[2] else None
        db_ih = dgx.sum(0) if need[3] else None
        db_hh = dgh.sum(0) if need[4] else None
        return dx, dw_ih, dw_hh, db_ih, db_hh, None


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding's lookup whose backward is, INSIDE a HIP-graph capture, ONE index_add_ (atomic adds into the zeroed table).
    torch's dense embedding backward switches, above 3072 indices, to a sort / unique-by-key path: ~10 launches and 70-180 us of
    host time per call where this one takes 10 (docs/history/tools/embed_sync_probe.py; neither waits for the device), and on this stack that
    path does not survive a HIP-graph capture -- the replay of a text client's step padded to 128 x 32 = 4096 indices died with a
    memory fault (profiles/r5_embed_backward_probe.json), the same step with this backward replays.
    The atomic adds make the fp32 sum order-dependent, so the index_add_ form is used ONLY while a stream is capturing (or with
    EMBED_INDEX_ADD[0] set: measurements); an eager step -- the replicated multi-rank server phases that are documented as bit-for-bit
    equal on every rank, and anything under torch.use_deterministic_algorithms(True) -- runs torch's own deterministic dense backward,
    i.e. exactly what the reference's nn.Embedding does.  A REPLAYED text / multi-modal client step is therefore deterministic only up
    to fp32 summation order in its embedding gradient (rows of repeated words)."""

    @staticmethod
    def forward(ctx, tokens, weight):
        ctx.save_for_backward(tokens)
        ctx.table = weight.shape
        return weight.index_select(0, tokens.reshape(-1)).view(*tokens.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        if not (EMBED_INDEX_ADD[0] or torch.cuda.is_current_stream_capturing()):
            return None, torch.ops.aten.embedding_dense_backward(g.contiguous(), tokens, ctx.table[0], -1, False)
        dw = torch.zeros(ctx.table, dtype=g.dtype, device=g.device)
        dw.index_add_(0, tokens.reshape(-1), g.reshape(-1, g.shape[-1]))
        return None, dw


EMBED_INDEX_ADD = [_os.environ.get('CFL_EMBED_INDEX_ADD', '0') == '1']


def embedding_lookup(embed, tokens):
    """`embed(tokens)` for a plain nn.Embedding on the GPU (no padding_idx / max_norm / sparse gradients / frequency scaling: the
    reference's text towers, language_model.py:39, caption_encoder.py:39) with the capturable backward above; anything else goes to
    the module itself."""
    w = embed.weight
    if (w.is_cuda and tokens.is_cuda and embed.padding_idx is None and embed.max_norm is None and not embed.sparse
            and not embed.scale_grad_by_freq and tokens.dtype == torch.int64):
        return _EmbeddingFn.apply(tokens, w)
    return embed(tokens)


def gru_last_supported(rnn, words=None):
    """True when `bigru_last_states` covers this nn.GRU (and, when given, this input): one bidirectional batch_first layer with
    biases, fp32 on the GPU, hidden width built into gru.hip; anything else stays on the library's GRU."""
    if not (GRU_FUSED[0] and isinstance(rnn, torch.nn.GRU) and rnn.num_layers == 1 and rnn.bidirectional and rnn.batch_first
            and rnn.bias and float(rnn.dropout) == 0.0 and rnn.weight_hh_l0.is_cuda and rnn.weight_hh_l0.dtype == torch.float32
            and not torch.is_autocast_enabled()):
        return False
    if words is not None and not (words.is_cuda and words.dtype == torch.float32 and words.dim() == 3):
        return False
    return bool(_lib.load().cfl_gru_supported(rnn.hidden_size))


def bigru_last_states(rnn, words, lengths):
    """`pad_packed_sequence(rnn(pack_padded_sequence(words, lengths)))[0].gather(1, lengths - 1)` of the reference's text towers
    (language_model.py:93-107, caption_encoder.py:87-101) without the packing and without the rest of the output: [B, 2H] =
    [forward direction after lengths[b] steps | backward direction's cell on word lengths[b] - 1].  `lengths` may live on the
    host (then it is checked like pack_padded_sequence checks it: sorted, positive) or on the device (no host round trip: the
    form the captured client step uses)."""
    B, T, _ = words.shape
    if not lengths.is_cuda:
        ll = lengths.tolist()
        if any(v <= 0 for v in ll):
            raise RuntimeError('Length of all samples has to be greater than 0, but found an element in \'lengths\' that is <= 0')
        if any(a < b for a, b in zip(ll, ll[1:])):
            raise RuntimeError('`lengths` array must be sorted in decreasing order when `enforce_sorted` is True.')
        if ll and ll[0] > T:
            raise RuntimeError(f'length {ll[0]} exceeds the padded width {T}')
    lens = lengths.to(device=words.device, dtype=torch.int32, non_blocking=True).contiguous()
    words = words.contiguous()
    track = torch.is_grad_enabled()
    fwd = _GruLastFn.apply(words, lens, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, track)
    last = (lens.to(torch.int64) - 1).clamp_(0, T - 1)
    x_last = words.gather(1, last.view(B, 1, 1).expand(B, 1, words.shape[2])).squeeze(1)
    bwd = _GruCell0Fn.apply(x_last, rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, rnn.bias_ih_l0_reverse,
                            rnn.bias_hh_l0_reverse, track)
    return torch.cat([fwd, bwd], 1)


class _SupGlueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fvec, labels, class_weight, margin, topk, center_weight):
        lib = _lib.load()
        B, C = fvec.shape
        Dw = class_weight.shape[1]
        out5 = torch.empty(5, dtype=torch.float32, device=fvec.device)
        ws = _ws(lib.cfl_sup_ws_bytes(B, C), fvec.device)
        _lib.check(lib.cfl_sup_glue_fwd(_ptr(fvec), _ptr(labels), _ptr(class_weight), B, C, Dw, margin, topk,
                                        center_weight, _ptr(out5), _ptr(ws), _stream(fvec)), 'cfl_sup_glue_fwd')
        ctx.save_for_backward(fvec, labels, class_weight, ws)
        ctx.hyper = (margin, center_weight)
        ctx.mark_non_differentiable(out5)
        return out5[0].clone(), out5

    @staticmethod
    def backward(ctx, g, _gstats):
        lib = _lib.load()
        fvec, labels, class_weight, ws = ctx.saved_tensors
        margin, center_weight = ctx.hyper
        B, C = fvec.shape
        Dw = class_weight.shape[1]
        dF = torch.empty_like(fvec) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(class_weight) if ctx.needs_input_grad[2] else None
        gg = g.reshape(1).to(torch.float32).contiguous()
        _lib.check(lib.cfl_sup_glue_bwd(_ptr(fvec), _ptr(labels), _ptr(class_weight), B, C, Dw, margin, center_weight,
                                        _ptr(gg), _ptr(ws), _ptr(dF), _ptr(dW), _stream(fvec)), 'cfl_sup_glue_bwd')
        return dF, None, dW, None, None, None


def supervised_glue(fvec, labels, class_weight, inter_distance, topk=5, center_weight=0.5):
    """SURVEY 8f-4, the client's supervised loss glue (ClientTrainer.py:344-361): one-hot margin subtract, CE,
    centre loss CE(class_weight @ class_weight^T, arange(C)) and precision@1 / @topk in three launches.
    Returns (total_loss 0-d with grad to fvec and class_weight, stats[5] = {total, ce, center, prec@1 %, prec@topk %})."""
    F = _f32(fvec, 'fvec')
    W = _f32(class_weight, 'class_weight')
    if F.dim() != 2 or W.dim() != 2 or W.shape[0] != F.shape[1]:
        raise RuntimeError(f'shape mismatch: fvec {tuple(F.shape)} class_weight {tuple(W.shape)}')
    y = _idx(labels, F.device)
    if y.numel() != F.shape[0]:
        raise RuntimeError(f'labels {tuple(y.shape)} do not match fvec {tuple(F.shape)}')
    return _SupGlueFn.apply(F, y, W, float(inter_distance), int(topk), float(center_weight))


# --------------------------------------------------------------------------- A5: con_w
@torch.no_grad()
def conw_logprob(vec, global_other, row0=0, rows=None):
    """A5 (MMFL.py:304-307) for rows [row0, row0+rows) of one client representation [M, D]."""
    lib = _lib.load()
    V = _f32(vec, 'vec')
    G = _f32(global_other, 'global_other')
    M, D = V.shape
    if G.shape != V.shape:
        raise RuntimeError(f'shape mismatch {tuple(V.shape)} vs {tuple(G.shape)}')
    rows = M - row0 if rows is None else rows
    out = torch.empty(rows, dtype=torch.float32, device=V.device)
    if (not _CONW_NOIMG and lib.cfl_conw_img_supported(rows, M, D) and V.data_ptr() % 16 == 0 and G.data_ptr() % 16 == 0
            and not (_BANK_EXACT or lib.cfl_get_exact_gemm())):
        # the bank pass of rows A3/A4 on the (cached) pre-split image of G: the bank moves through a CU once per 256 rows (128 beyond D = 256)
        img = bank_image(G)
        ws = _ws(lib.cfl_conw_img_ws_bytes(rows, M, D), V.device)
        _lib.check(lib.cfl_conw_logprob_img(_ptr(V), img.data_ptr(), _ptr(G), M, D, row0, rows, _ptr(out), _ptr(ws), _stream(V)),
                   'cfl_conw_logprob_img')
        return out
    ws = _ws(lib.cfl_conw_ws_bytes(rows, M, D), V.device)
    _lib.check(lib.cfl_conw_logprob(_ptr(V), _ptr(G), M, D, row0, rows, _ptr(out), _ptr(ws), _stream(V)),
               'cfl_conw_logprob')
    return out


@torch.no_grad()
def conw_combine(vecs, logprobs, return_weights=False):
    """A5 (MMFL.py:311-314): softmax over clients of logprobs [C, M], weighted sum of vecs."""
    lib = _lib.load()
    vecs = [_f32(v, 'vec') for v in vecs]
    L = _f32(logprobs, 'logprobs')
    C = len(vecs)
    M, D = vecs[0].shape
    if L.shape != (C, M):
        raise RuntimeError(f'logprobs shape {tuple(L.shape)} != ({C}, {M})')
    out = torch.empty(M, D, dtype=torch.float32, device=L.device)
    W = torch.empty(C, M, dtype=torch.float32, device=L.device) if return_weights else None
    arr = (ctypes.c_void_p * C)(*[v.data_ptr() for v in vecs])
    _lib.check(lib.cfl_conw_combine(arr, _ptr(L), C, M, D, _ptr(out), _ptr(W), _stream(L)), 'cfl_conw_combine')
    return (out, W) if return_weights else out


# --------------------------------------------------------------------------- A2-head: PIE
class _PiePoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, H, w2, mask, want_mean):
        lib = _lib.load()
        N, P, Cd = X.shape
        dh = H.shape[2]
        dev = X.device
        attn = torch.empty(N, P, dtype=torch.float32, device=dev)
        pooled = torch.empty(N, Cd, dtype=torch.float32, device=dev)
        xmean = torch.empty(N, Cd, dtype=torch.float32, device=dev) if want_mean else None
        ws = _ws(lib.cfl_pie_ws_bytes(N, P, Cd, dh), dev)
        _lib.check(lib.cfl_pie_pool_fwd(_ptr(X), _ptr(H), _ptr(w2), _ptr(mask), N, P, Cd, dh, _ptr(attn), _ptr(pooled),
                                        _ptr(xmean), _ptr(ws), _stream(X)), 'cfl_pie_pool_fwd')
        ctx.save_for_backward(X, H, w2, attn)
        ctx.mask = mask
        ctx.want_mean = want_mean
        ctx.mark_non_differentiable(attn)
        if want_mean:
            return pooled, attn, xmean
        return pooled, attn, pooled.new_empty(0)

    @staticmethod
    def backward(ctx, dpooled, _dattn, dxmean):
        lib = _lib.load()
        X, H, w2, attn = ctx.saved_tensors
        N, P, Cd = X.shape
        dh = H.shape[2]
        dpooled = dpooled.contiguous().float()
        dxm = dxmean.contiguous().float() if ctx.want_mean else None
        dX = torch.empty_like(X)
        dH = torch.empty_like(H)
        dw2 = torch.empty_like(w2)
        ws = _ws(lib.cfl_pie_ws_bytes(N, P, Cd, dh), X.device)
        _lib.check(lib.cfl_pie_pool_bwd(_ptr(X), _ptr(H), _ptr(w2), _ptr(ctx.mask), _ptr(attn), _ptr(dpooled), _ptr(dxm),
                                        N, P, Cd, dh, _ptr(dX), _ptr(dH), _ptr(dw2), _ptr(ws), _stream(X)),
                   'cfl_pie_pool_bwd')
        return dX, dH, dw2, None, None


class _PieHeadFn(torch.autograd.Function):
    """Single-pass attention pooling (csrc/pie_fused.hip): X / H in the dtype they were produced in (fp32 or bf16)."""

    @staticmethod
    def forward(ctx, X, H, w2, mask, want_mean):
        lib = _lib.load()
        N, P, Cd = X.shape
        dh = H.shape[2]
        dev = X.device
        bf16 = int(X.dtype == torch.bfloat16)
        attn = torch.empty(N, P, dtype=torch.float32, device=dev)
        pooled = torch.empty(N, Cd, dtype=torch.float32, device=dev)
        xmean = torch.empty(N, Cd, dtype=torch.float32, device=dev) if want_mean else None
        _lib.check(lib.cfl_pie_head_fwd(_ptr(X), _ptr(H), bf16, _ptr(w2), _ptr(mask), N, P, Cd, dh, _ptr(attn), _ptr(pooled),
                                        _ptr(xmean), _stream(X)), 'cfl_pie_head_fwd')
        ctx.save_for_backward(X, H, w2, attn)
        ctx.want_mean = want_mean
        ctx.mark_non_differentiable(attn)
        if want_mean:
            return pooled, attn, xmean
        return pooled, attn, pooled.new_empty(0)

    @staticmethod
    def backward(ctx, dpooled, _dattn, dxmean):
        lib = _lib.load()
        X, H, w2, attn = ctx.saved_tensors
        N, P, Cd = X.shape
        dh = H.shape[2]
        dpooled = dpooled.contiguous().float()
        dxm = dxmean.contiguous().float() if ctx.want_mean else None
        dX = torch.empty_like(X)
        dH = torch.empty_like(H)
        dw2 = torch.empty_like(w2)
        ws = _ws(lib.cfl_pie_ws_bytes(N, P, Cd, dh), X.device)
        _lib.check(lib.cfl_pie_head_bwd(_ptr(X), _ptr(H), int(X.dtype == torch.bfloat16), _ptr(w2), _ptr(attn), _ptr(dpooled),
                                        _ptr(dxm), N, P, Cd, dh, _ptr(dX), _ptr(dH), _ptr(dw2), _ptr(ws), _stream(X)),
                   'cfl_pie_head_bwd')
        return dX, dH, dw2, None, None


PIE_FUSED = _os.environ.get('CFL_PIE_UNFUSED', '0') != '1'      # tests flip this to compare the two implementations


def pie_pool(x, h, w2, pad_mask=None, want_mean=False):
    """softmax_P(w2 . tanh(h)) attention pooling of x (pie_model.py:28-40, n_head = 1).
    x [N,P,Cd], h = w_1(x) [N,P,dh], w2 [dh] or [1,dh], pad_mask [N,P] bool (True = padded).
    Returns (pooled [N,Cd], attn [N,P], xmean [N,Cd] or empty).
    fp32 or bf16 x / h (the autocast regime hands over bf16) go through the single-pass kernels in that dtype when the
    shape allows (cfl_pie_fused_supported); anything else is converted to fp32 for the three-pass kernels of pie.hip."""
    if not (torch.is_tensor(x) and x.is_cuda and torch.is_tensor(h) and h.is_cuda):
        raise _lib.CreamflHipError('pie_pool: expected CUDA/HIP tensors (no CPU fallback in creamfl_amd)')
    W2 = _f32(w2, 'w2').reshape(-1)
    if x.dim() != 3 or h.dim() != 3 or x.shape[:2] != h.shape[:2] or W2.numel() != h.shape[2]:
        raise RuntimeError(f'pie_pool shape mismatch x{tuple(x.shape)} h{tuple(h.shape)} w2{tuple(W2.shape)}')
    m = None
    if pad_mask is not None:
        m = pad_mask.to(device=x.device, dtype=torch.uint8).contiguous()
    N, P, Cd = x.shape
    if (PIE_FUSED and x.dtype == h.dtype and x.dtype in (torch.float32, torch.bfloat16)
            and _lib.load().cfl_pie_fused_supported(N, P, Cd, h.shape[2], int(x.dtype == torch.bfloat16))):
        return _PieHeadFn.apply(x.contiguous(), h.contiguous(), W2, m, bool(want_mean))
    return _PiePoolFn.apply(_f32(x, 'x'), _f32(h, 'h'), W2, m, bool(want_mean))


class _PieEpilogueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, res_pre, ln_w, ln_b, eps, flags):
        lib = _lib.load()
        N, D = out.shape
        dev = out.device
        y = torch.empty_like(out)
        o = torch.empty_like(out)
        r = torch.empty_like(out)
        stats = torch.empty(N, 4, dtype=torch.float32, device=dev)
        _lib.check(lib.cfl_pie_epilogue_fwd(_ptr(out), _ptr(res_pre), _ptr(ln_w), _ptr(ln_b), N, D, eps, flags, _ptr(y),
                                            _ptr(o), _ptr(r), _ptr(stats), _stream(out)), 'cfl_pie_epilogue_fwd')
        ctx.save_for_backward(out, r, ln_w, ln_b, stats)
        ctx.flags = flags
        ctx.set_materialize_grads(False)
        return y, o, r

    @staticmethod
    def backward(ctx, dy, do_, dres):
        lib = _lib.load()
        out, r, ln_w, ln_b, stats = ctx.saved_tensors
        N, D = out.shape
        dy = dy.contiguous().float() if dy is not None else torch.zeros_like(out)
        do_ = do_.contiguous().float() if do_ is not None else None
        dres = dres.contiguous().float() if dres is not None else None
        d_out = torch.empty_like(out)
        d_res = torch.empty_like(out)
        d_w = torch.empty_like(ln_w)
        d_b = torch.empty_like(ln_b)
        ws = _ws(lib.cfl_pie_ws_bytes(N, 1, D, 1), out.device)
        _lib.check(lib.cfl_pie_epilogue_bwd(_ptr(dy), _ptr(do_), _ptr(dres), _ptr(out), _ptr(r), _ptr(ln_w), _ptr(ln_b),
                                            _ptr(stats), N, D, ctx.flags, _ptr(d_out), _ptr(d_res), _ptr(d_w), _ptr(d_b),
                                            _ptr(ws), _stream(out)), 'cfl_pie_epilogue_bwd')
        return d_out, d_res, d_w, d_b, None, None


def pie_epilogue(out, res_pre, ln_w, ln_b, eps=1e-5, l2norm=True):
    """r = sigmoid(res_pre); o = LayerNorm(out + r); y = l2_normalize(o) (pie_model.py:63-66 +
    tensor_utils.py:25-27).  Returns (y, o, r); with l2norm=False, y == o."""
    return _PieEpilogueFn.apply(_f32(out, 'out'), _f32(res_pre, 'res_pre'), _f32(ln_w, 'ln_w'), _f32(ln_b, 'ln_b'),
                                float(eps), 0 if l2norm else 1)


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        N, D = x.shape
        y = torch.empty_like(x)
        inv = torch.empty(N, dtype=torch.float32, device=x.device)
        _lib.check(lib.cfl_l2norm_fwd(_ptr(x), N, D, _ptr(y), _ptr(inv), _stream(x)), 'cfl_l2norm_fwd')
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        y, inv = ctx.saved_tensors
        N, D = y.shape
        dy = dy.contiguous().float()
        dx = torch.empty_like(y)
        _lib.check(lib.cfl_l2norm_bwd(_ptr(dy), _ptr(y), _ptr(inv), N, D, _ptr(dx), _stream(y)), 'cfl_l2norm_bwd')
        return dx


def l2_normalize(tensor, axis=-1):
    """src/utils/tensor_utils.py:25-27 for 2-D [N, D] tensors normalised along the last axis."""
    x = _f32(tensor, 'tensor')
    if x.dim() != 2 or axis not in (-1, 1):
        raise RuntimeError('creamfl_amd.ops.l2_normalize handles [N, D] tensors along the last axis')
    return _L2NormFn.apply(x)


# --------------------------------------------------------------------------- trunk glue: fused BN (+add) (+ReLU)
BN_FP32 = [_os.environ.get('CFL_NO_BN_FP32', '0') != '1']      # switch: fp32 channels_last activations on the fused kernels too


def bn_act_supported(x, num_features):
    """True when the fused NHWC BatchNorm kernels apply to `x`: [N, C, H, W] channels_last, bf16 (the server's trunks) or -- round
    5, the clients' fp32 encoders -- fp32 (the same kernels instantiated on 32-byte channel groups: cfl_bn_*_f32)."""
    c8 = num_features // 8
    return (x.is_cuda and (x.dtype == torch.bfloat16 or (x.dtype == torch.float32 and BN_FP32[0])) and x.dim() == 4
            and num_features % 8 == 0
            and (256 % c8 == 0 if c8 < 256 else num_features % 2048 == 0)
            and x.is_contiguous(memory_format=torch.channels_last))


# ---- gradient join of a residual block fused into the 1x1 data-gradient GEMM (round 3) --------------------------------------
# z = relu(bn3(c3) + r) feeds the next block twice: its first 1x1 convolution and its skip connection.  Backward, bn3 needs
# g = (A + B) . mask with A = that convolution's data gradient (the hand-written GEMM) and B = the gradient arriving over the
# skip connection (the next block's bn3 hands on its own masked gradient).  Unfused, BOTH BatchNorm backward passes read A, B
# and the mask, and the apply pass writes g again as the residual gradient.  Fused: the GEMM epilogue reads B and the mask and
# writes g; the BatchNorm backward reads g alone (no mask, no second gradient) and hands the same tensor on: -4 bytes per
# element of every bn3 output with a direct skip connection (29 of ResNet-101's 33 blocks).  The pieces meet through
# these registries, keyed by a TOKEN: a serial number bn_act_train gives every output while the registries are armed and
# attaches to the two tensor objects it returns (`_cfl_tok`); the consuming convolution and the next block's BatchNorm read it
# off the tensor OBJECT they are handed.  (Not the buffer address: a downsample branch's output dies inside the forward pass, the
# allocator hands its address to a later block's output, and an address key then diverts the downsample gradient -- seen as
# BatchNorm parameters without a gradient in some steps.)  The registries are only consulted inside TrainerEngine.backward
# (prepare_ / release_weight_transposes bracket it and clear them), so client trainers and plain autograd are untouched.
_NO_JOIN_FUSE = _os.environ.get('CFL_NO_JOIN_FUSE', '0') == '1'      # measurement switch
# Structural assumption, CHECKED in the BatchNorm backward: the pre-joined output object `y` is consumed by the block's first 1x1
# convolution ONLY (and its alias y2 by the next bn3's residual input only).  JOIN['pre'][tok] records the address of the g the
# GEMM wrote (and holds the tensor, so that autograd cannot accumulate another gradient into it in place); the BatchNorm backward requires the gradient it is handed to BE that tensor -- if anything else consumed y (a
# feature tap, a hook, another block variant) autograd has added an UNMASKED gradient into a new tensor, and the layer raises
# instead of back-propagating a silently wrong sum.
JOIN = {'armed': False, 'on': False, 'serial': 0, 'mask': {}, 'consumer': set(), 'pending': {}, 'pre': {}, 'fused': 0}


def join_arm():
    """From here on the fused BatchNorm layers register their ReLU masks and the 1x1 convolutions themselves as consumers.
    Use through `join_scope()` (TrainerEngine.train_step, the KD step): a bare call leaves the registries armed -- and the mask
    tensors referenced -- until the next TrainerEngine.backward."""
    JOIN['armed'] = not _NO_JOIN_FUSE
    JOIN['mask'].clear()
    JOIN['consumer'].clear()


@contextlib.contextmanager
def join_scope():
    """Arms the gradient-join registries for ONE step -- the forward pass and the TrainerEngine.backward inside the `with` -- and
    disarms / empties them on the way out, exception or not: no stale armed state, no mask tensors kept alive past the step, and
    forwards of other models in the process (client trainers, evaluation) never register anything."""
    join_arm()
    try:
        yield
    finally:
        _join_reset(False)


def _join_reset(on):
    JOIN['on'] = bool(on) and JOIN['armed']
    JOIN['pending'].clear()
    JOIN['pre'].clear()
    if not on:
        JOIN['armed'] = False
        JOIN['mask'].clear()
        JOIN['consumer'].clear()


# element counters of the fused BN kernels (python ints; bench.py turns them into algorithmic bytes)
BN_COUNTERS = {'fwd': 0, 'fwd_pre': 0, 'fwd_res': 0, 'fwd_mask': 0, 'bwd': 0, 'bwd_relu': 0, 'bwd_res': 0, 'bwd_two': 0, 'bwd_wg': 0}


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, tok=0, res_tok=0, pstat=None,
                pstat_nblk=0, wg=None):
        lib = _lib.load()
        N, C, H, W = x.shape
        R = N * H * W
        BN_COUNTERS['fwd'] += R * C
        if residual is not None:
            BN_COUNTERS['fwd_res'] += R * C
        y = torch.empty_like(x)                           # keeps channels_last
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _ws(lib.cfl_bn_ws_bytes(R, C), x.device)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        # ReLU mask for the backward: without a residual it is recomputed from x, gamma, beta; with one the forward
        # packs it into 1 bit per element (R*C/8 bytes) so that the backward does not re-read y (16x fewer bytes, twice)
        need_mask = ctx.relu and ctx.has_res and any(ctx.needs_input_grad[:4])
        mask = torch.empty(R * C // 8, dtype=torch.uint8, device=x.device) if need_mask else None
        if need_mask:
            BN_COUNTERS['fwd_mask'] += R * C
        if pstat is not None:        # the producing GEMM left the per-column partial sums: no statistics pass
            BN_COUNTERS['fwd_pre'] += R * C
            _lib.check(lib.cfl_bn_fwd_pre(_ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                                          R, C, eps, momentum, int(relu), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(mask),
                                          ctypes.c_void_p(pstat.data_ptr()), ctypes.c_void_p(pstat.data_ptr() + 4 * pstat_nblk * C),
                                          pstat_nblk, _stream(x)), 'cfl_bn_fwd_pre')
        else:
            _lib.check((lib.cfl_bn_fwd_f32 if x.dtype == torch.float32 else lib.cfl_bn_fwd)(
                _ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                                      R, C, eps, momentum, int(relu), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(mask), _ptr(ws),
                                      _stream(x)), 'cfl_bn_fwd')
        ctx.save_for_backward(x, mask if need_mask else x, weight, bias, mean, invstd)
        ctx.set_materialize_grads(False)                  # an unused alias must arrive as None, not as a zero tensor
        ctx.wg = wg                                       # weight-gradient holder of the convolution that made x (WGRAD_FUSE)
        ctx.tok = tok                                     # join tokens of the output / of the residual input (0 = none)
        ctx.res_tok = res_tok if residual is not None else 0
        if need_mask and JOIN['armed'] and tok:
            JOIN['mask'][tok] = mask                     # the consumer GEMM of y applies it (see JOIN above)
        return y, _alias(y)

    @staticmethod
    def backward(ctx, dy, dy2):
        lib = _lib.load()
        x, mask, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        R = N * H * W
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return (None,) * 14
        # JOIN: the data-gradient GEMM of the consumer already produced g = (A + B) . mask for this layer's output
        pre = bool(JOIN['on'] and ctx.tok and ctx.tok in JOIN['pre'])
        if pre:
            joined_at, _held = JOIN['pre'].pop(ctx.tok)
            del _held
            if dy2 is not None or dy.data_ptr() != joined_at:
                raise _lib.CreamflHipError(
                    'fused gradient join: the output of a pre-joined BatchNorm layer was consumed by something besides the '
                    "block's first 1x1 convolution (a second gradient, or a sum autograd built from an unmasked one, reached it); "
                    'run this model with CFL_NO_JOIN_FUSE=1')
        from . import streams
        if streams.FLUSH_POLICY[0] == 2:
            streams.flush(x.device, 1)
        elif streams.FLUSH_POLICY[0] >= 3:
            if ctx.has_res and len(streams._PENDING) >= streams.FLUSH_POLICY[0]:
                streams.flush(x.device)
        elif ctx.has_res or streams.FLUSH_POLICY[0] == 1:
            streams.flush(x.device)          # a long HBM-bound phase starts: let the queued weight gradients run beside it
        BN_COUNTERS['bwd'] += R * C
        if ctx.relu and ctx.has_res and not pre:
            BN_COUNTERS['bwd_relu'] += R * C              # passes that read the 1-bit ReLU mask
        if ctx.has_res and not pre:
            BN_COUNTERS['bwd_res'] += R * C
        if dy2 is not None:
            BN_COUNTERS['bwd_two'] += R * C

        bn_bwd = lib.cfl_bn_bwd_f32 if x.dtype == torch.float32 else lib.cfl_bn_bwd

        def prep(t):
            if t.dtype != x.dtype:
                t = t.to(x.dtype)
            return t.contiguous(memory_format=torch.channels_last)
        dy = prep(dy)
        dy2 = prep(dy2) if dy2 is not None else None
        dx = torch.empty_like(x)
        dres = dy if pre else (torch.empty_like(x) if ctx.has_res else None)      # pre-joined: g IS the residual gradient
        dgamma = torch.empty_like(weight)
        dbeta = torch.empty_like(weight)
        ws = _ws(lib.cfl_bn_ws_bytes(R, C), x.device)
        wg = ctx.wg
        if pre and wg is not None and WGRAD_FUSE[0] and x.dtype == torch.bfloat16 and wg['x'] is not None and not wg['done'] and \
                lib.cfl_bn_bwd_wgrad_supported(R, C, wg['w'].shape[1]) and wg['x'].is_contiguous(memory_format=torch.channels_last):
            # ... and the weight gradient of the convolution that made x, from the dY tile while it is on the chip
            w = wg['w']
            P = w.shape[1]
            dw = torch.empty_like(w)
            ws2 = _ws(lib.cfl_bn_bwd_wgrad_ws_bytes(R, C, P), x.device)
            _lib.check(lib.cfl_bn_bwd_wgrad(_ptr(dy), _ptr(x), _ptr(wg['x']), P, _ptr(weight), _ptr(mean), _ptr(invstd), R, C, _ptr(dx),
                                            _ptr(dgamma), _ptr(dbeta), _ptr(dw), _ptr(ws2), _stream(x)), 'cfl_bn_bwd_wgrad')
            with torch.no_grad():
                if w.grad is None:
                    w.grad = dw
                else:
                    w.grad.add_(dw)
            wg['done'] = True
            WGRAD_FUSED[0] += 1
            BN_COUNTERS['bwd'] -= R * C                   # (its apply pass is another kernel id: counted apart)
            BN_COUNTERS['bwd_wg'] += R * C
            from . import streams as _st
            if _st.GRAD_READY[0] is not None:
                _st.GRAD_READY[0](w)                      # multi-GPU: this gradient may now be bucketed
        elif pre:          # plain BatchNorm backward of an already masked, already summed gradient
            _lib.check(bn_bwd(_ptr(dy), _ptr(None), _ptr(x), _ptr(None), _ptr(None), _ptr(weight), _ptr(bias), _ptr(mean),
                                      _ptr(invstd), R, C, 0, 0, _ptr(dx), _ptr(None), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream(x)),
                       'cfl_bn_bwd')
        else:
            _lib.check(bn_bwd(_ptr(dy), _ptr(dy2), _ptr(x), _ptr(None), _ptr(mask) if (ctx.relu and ctx.has_res) else _ptr(None),
                                      _ptr(weight), _ptr(bias), _ptr(mean), _ptr(invstd), R, C, int(ctx.relu), int(ctx.has_res),
                                      _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream(x)), 'cfl_bn_bwd')
        if JOIN['on'] and ctx.has_res and ctx.res_tok and ctx.res_tok in JOIN['consumer'] and ctx.res_tok in JOIN['mask']:
            # the skip connection's gradient goes to the data-gradient GEMM of this block's first convolution instead of to
            # autograd (None = no contribution): that GEMM adds it and masks the sum for the BatchNorm below
            JOIN['pending'][ctx.res_tok] = dres
            dres = None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def bn_act_train(x, weight, bias, running_mean, running_var, momentum, eps, relu=False, residual=None, two=False):
    """Training-mode BatchNorm2d (+ residual) (+ ReLU) on a channels_last bf16 activation (csrc/bnorm.hip).
    `two=True` returns the output as two tensor objects on one buffer: give one to the next convolution and the other
    to the next residual add, and their two gradients are summed inside the fused backward (no autograd add kernel)."""
    res_tok = getattr(residual, '_cfl_tok', 0) if residual is not None else 0
    if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype and bn_act_supported(residual, x.shape[1])):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
        res_tok = 0
    tok = 0
    if JOIN['armed']:
        JOIN['serial'] += 1
        tok = JOIN['serial']
    pre = getattr(x, '_cfl_bnstats', None) if x.dtype == torch.bfloat16 else None
    if pre is not None and not (CONV_STATS[0] and pre[2] == x.shape[0] * x.shape[2] * x.shape[3] and pre[3] == x.shape[1]):
        pre = None
    y, y2 = _BNActFn.apply(x, residual, weight, bias, running_mean, running_var, float(momentum), float(eps), bool(relu), tok, res_tok,
                           pre[0] if pre is not None else None, pre[1] if pre is not None else 0, getattr(x, '_cfl_wg', None))
    if tok:
        y._cfl_tok = tok
        y2._cfl_tok = tok
    return (y, y2) if two else y


class _BnReluMaxPoolFn(torch.autograd.Function):
    """ResNet stem tail: BatchNorm (batch statistics) + ReLU + MaxPool2d(3, 2, 1) without the normalised activation or the
    scattered pooling gradient ever reaching memory (csrc/bnorm.hip: cfl_bn_pool_fwd / cfl_bn_pool_bwd); results are
    bit-identical to bn_act_train(relu=True) followed by maxpool3s2."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        lib = _lib.load()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty(N * Ho * Wo * C, dtype=torch.uint8, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _ws(lib.cfl_bn_ws_bytes(N * H * W, C), x.device)
        fn = lib.cfl_bn_pool_fwd_f32 if x.dtype == torch.float32 else lib.cfl_bn_pool_fwd
        _lib.check(fn(_ptr(x), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var), N, H, W, C, eps,
                      momentum, _ptr(y), _ptr(idx), _ptr(mean), _ptr(invstd), _ptr(ws), _stream(x)), 'cfl_bn_pool_fwd')
        ctx.save_for_backward(x, idx, weight, bias, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, idx, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gy = gy.contiguous(memory_format=torch.channels_last)
        from . import streams
        streams.flush(x.device)                # a long HBM-bound phase: let the queued weight gradients run beside it
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(weight)
        dbeta = torch.empty_like(weight)
        ws = _ws(lib.cfl_bn_ws_bytes(N * H * W, C), x.device)
        fn = lib.cfl_bn_pool_bwd_f32 if x.dtype == torch.float32 else lib.cfl_bn_pool_bwd
        _lib.check(fn(_ptr(gy), _ptr(idx), _ptr(x), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(invstd), N, H, W, C,
                      _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream(x)), 'cfl_bn_pool_bwd')
        return dx, dgamma, dbeta, None, None, None, None


_NO_STEM_TAIL = _os.environ.get('CFL_NO_STEM_TAIL', '0') == '1'     # measurement switch: BatchNorm and pooling as two ops


def bn_relu_maxpool_supported(x, num_features):
    # (fp32: the clients' encoders, round 6 -- the same kernels on 32-byte channel groups)
    return (not _NO_STEM_TAIL and x.dtype in (torch.bfloat16, torch.float32) and bn_act_supported(x, num_features) and num_features <= 2048
            and x.shape[2] >= 2 and x.shape[3] >= 2 and torch.is_grad_enabled())


def bn_relu_maxpool(x, weight, bias, running_mean, running_var, momentum, eps):
    return _BnReluMaxPoolFn.apply(x, weight, bias, running_mean, running_var, float(momentum), float(eps))


@torch.no_grad()
def bn_act_eval(x, weight, bias, running_mean, running_var, eps, relu=False, residual=None):
    """Evaluation-mode BatchNorm2d (+ residual) (+ ReLU), no autograd."""
    lib = _lib.load()
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    invstd = torch.rsqrt(running_var.float() + eps)
    if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype and bn_act_supported(residual, C)):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    _lib.check((lib.cfl_bn_apply_f32 if x.dtype == torch.float32 else lib.cfl_bn_apply)(_ptr(x), _ptr(residual), _ptr(running_mean), _ptr(invstd), _ptr(weight), _ptr(bias),
                                N * H * W, C, int(relu), _ptr(y), _stream(x)), 'cfl_bn_apply')
    return y


# --------------------------------------------------------------------------- 1x1 convolution: data gradient (csrc/gemm_bf16.hip)
GEMM_COUNTERS = {'flops': 0, 'bytes': 0}   # python ints; bench.py turns them into algorithmic work per launch


def gemm_bf16_nt(a, b, out=None, variant=0, add=None, mask=None):
    """out[M, N] = a[M, K] @ b[N, K]^T, bf16 (csrc/gemm_bf16.hip: cfl_gemm_bf16_nt).  With `add` (bf16, the dense [M, N]
    matrix in out's memory order) and `mask` (1 bit per element, cfl_bn_fwd's packing): out = (a b^T + add) . mask."""
    lib = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    GEMM_COUNTERS['flops'] += 2 * M * N * K
    GEMM_COUNTERS['bytes'] += 2 * (M * K + N * K + M * N)
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    if add is not None:
        if out.stride(0) != N or add.numel() != M * N or add.dtype != torch.bfloat16 or mask.numel() * 8 != M * N:
            raise _lib.CreamflHipError('gemm_bf16_nt(add=, mask=): dense bf16 [M, N] operands and an M N / 8 byte mask expected')
        GEMM_COUNTERS['bytes'] += 2 * M * N + M * N // 8
        _lib.check(lib.cfl_gemm_bf16_nt_join(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), _ptr(add), _ptr(mask), M, N, K,
                                             _stream(a)), 'cfl_gemm_bf16_nt_join')
        return out
    _lib.check(lib.cfl_gemm_bf16_nt(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K, variant,
                                    _stream(a)), 'cfl_gemm_bf16_nt')
    return out


# ---- all weight transposes of a backward pass in one launch -----------------------------------------------------------
_WT = {'key': None, 'meta': None, 'flat': None, 'views': {}, 'tiles': 0, 'n': 0, 'valid': False}


def _rot_ok(w):
    """k x k (odd k > 1) bf16 weight in channels_last memory ([Co][k][k][Ci]): its data gradient can run on the FORWARD
    convolution kernel with the rotated, transposed weight."""
    return (w.dim() == 4 and w.shape[2] == w.shape[3] and w.shape[2] > 1 and w.shape[2] % 2 == 1
            and w.is_contiguous(memory_format=torch.channels_last))


def prepare_weight_transposes(weights):
    """ONE kernel launch for every weight transform the data gradients of the backward pass that follows need:
      * 1x1 weights [Co, Ci, 1, 1] -> W^T [Ci, Co] for the data-gradient GEMM (otherwise each GEMM launches its own small
        transpose on the critical path);
      * k x k weights (odd k, channels_last) -> W'[ci, co, kh, kw] = W[co, ci, k-1-kh, k-1-kw], with which the stride-1 data
        gradient is the FORWARD convolution of dy (_ConvSplitFn.backward).
    Call right before `loss.backward()`; `release_weight_transposes()` after it."""
    _join_reset(True)
    import numpy as np
    ws = [w for w in weights if w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 4
          and ((w.shape[2] == 1 and w.shape[3] == 1 and w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0) or _rot_ok(w))]
    if not ws:
        return
    lib = _lib.load()
    key = tuple((w.data_ptr(), tuple(w.shape)) for w in ws)
    if key != _WT['key']:
        dev = ws[0].device
        total = sum(w.numel() for w in ws)
        flat = torch.empty(total, dtype=torch.bfloat16, device=dev)
        rec = []
        views, off, tile0 = {}, 0, 0
        for w in ws:
            Co, Ci, k = w.shape[0], w.shape[1], w.shape[2]
            tc = (Ci + 63) // 64
            ntile = ((Co + 63) // 64) * tc
            if k == 1:
                v = flat[off:off + Co * Ci].view(Ci, Co)
                rec.append((w.data_ptr(), v.data_ptr(), Co, Ci, tile0, tc, Ci, Co))
                tile0 += ntile
            else:
                v = flat[off:off + w.numel()].view(Ci, k, k, Co).permute(0, 3, 1, 2)     # [Ci, Co, k, k], channels_last
                for kh in range(k):
                    for kw in range(k):
                        src = w.data_ptr() + ((k - 1 - kh) * k + (k - 1 - kw)) * Ci * 2
                        dst = v.data_ptr() + (kh * k + kw) * Co * 2
                        rec.append((src, dst, Co, Ci, tile0, tc, k * k * Ci, k * k * Co))
                        tile0 += ntile
            views[w.data_ptr()] = v
            off += w.numel()
        meta = np.array(rec, dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('R', '<i4'), ('C', '<i4'), ('tile0', '<i4'),
                                             ('tiles_c', '<i4'), ('lds', '<i4'), ('ldd', '<i4')]))
        _WT.update(key=key, flat=flat, views=views, tiles=tile0, n=len(rec),
                   meta=torch.from_numpy(meta.view(np.uint8).copy()).to(dev))
    _lib.check(lib.cfl_transpose_bf16_multi(_ptr(_WT['meta']), _WT['n'], _WT['tiles'], _stream(ws[0])), 'cfl_transpose_bf16_multi')
    _WT['valid'] = True


def release_weight_transposes():
    _WT['valid'] = False
    _join_reset(False)


def conv1x1_supported(x, weight):
    """1x1 / stride 1 convolution of a channels_last bf16 activation with a bf16 weight whose channel counts fit the
    GEMM kernel (Co % 64 == 0 for the reduction of the data gradient, Ci % 8 == 0)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and weight.shape[0] % 64 == 0 and weight.shape[1] % 8 == 0)


_JOIN_QUEUED_FOR = [-1]
DGRAD_PLAIN_LIB = [0]        # > 0: un-joined 1x1 data gradients with Co >= this run on the library (measurement knob)
_NO_STEM_S2D = _os.environ.get('CFL_NO_STEM_S2D', '0') == '1'        # measurement switch: the stem as the library sees it
_NO_FWD_DGRAD = _os.environ.get('CFL_NO_FWD_DGRAD', '0') == '1'      # measurement switch: MIOpen backward-data for k x k


def _queue_stream_join(device):
    """At the end of the running backward pass (once per pass: keyed by the autograd graph-task id, so an aborted
    pass cannot leave a stale flag behind) make the caller's stream wait for the auxiliary gradient streams."""
    gid = torch._C._current_graph_task_id()
    if gid >= 0 and _JOIN_QUEUED_FOR[0] == gid:
        return
    _JOIN_QUEUED_FOR[0] = gid

    def _join():
        _JOIN_QUEUED_FOR[0] = -1
        from . import streams
        streams.flush(device)
        streams.join_into_current(device)
    torch.autograd.Variable._execution_engine.queue_callback(_join)


# ---- MIOpen immediate mode where the recorded find-db holds the problem ---------------------------------------------------------
# PyTorch's `cudnn.benchmark = True` (creamfl_amd/runtime.py) makes MIOpen TIME its solvers for every new convolution problem of a
# process -- 56 s in the first server step of BASELINE configs[1] (12.6 s forward, 43.9 s backward), in EVERY process, although the
# find-db this package ships already holds the answers (PyTorch asks for an exhaustive search, which skips the db).  Immediate mode
# answers from the db without timing anything (first step < 1 s, same step time) but hands a shape the db does NOT hold a fallback
# kernel without saying so.  The trunk's convolution calls below therefore ask for immediate mode per call, and only when the
# problem's key is in the find-db this package ships and the process runs on (`Ci-H-W-kxk-Co-Ho-Wo-N-pad-stride-dilation-0-
# layouts-dtype-F`, backward problems written from the output side); every other problem keeps the timed search.  A key this code
# builds wrongly (another MIOpen version) is simply not found: slower start, never a slower kernel.  CFL_MIOPEN_AUTO=0 disables.
_FDB = {'keys': None, 'known': {}, 'hits': 0, 'misses': 0, 'on': _os.environ.get('CFL_MIOPEN_AUTO', '1') != '0'}


def _fdb_keys():
    """Keys of the find-db files THIS PACKAGE ships (creamfl_amd/miopen_db, seeded into the process's MIOPEN_USER_DB_PATH by
    runtime.configure_env) -- not of whatever the user directory has accumulated since: for the shipped problems immediate mode
    was measured against the timed search (same step time); a problem recorded later by some run has not been, and one such
    check (batch 512: 132 vs 86 ms per step) says it must not be assumed.  A shipped record counts only while the copy MIOpen
    answers from still holds the SAME record (a copy seeded by another package version, or a record a later timed search rewrote,
    is not what was measured: that problem goes back to the timed search)."""
    if _FDB['keys'] is None:
        keys = set()
        from . import runtime as _rt
        user = _os.environ.get('MIOPEN_USER_DB_PATH')
        try:
            if user and _os.environ.get('CFL_SEEDED_DB') == '1':          # the process really runs on (a copy of) the shipped db
                for fn in _os.listdir(_rt.DB_SRC):
                    if fn.endswith('.ufdb.txt') and _os.path.exists(_os.path.join(user, fn)):
                        with open(_os.path.join(user, fn)) as f:
                            live = {line.rstrip('\n') for line in f if '=' in line}
                        with open(_os.path.join(_rt.DB_SRC, fn)) as f:
                            for line in f:
                                if '=' in line and line.rstrip('\n') in live:
                                    keys.add(line.split('=', 1)[0])
        except OSError:
            pass
        _FDB['keys'] = keys
    return _FDB['keys']


def fdb_key(direction, xs, ws, os_, stride, padding, nhwc=True, dtype='BF16'):
    """The find-db key of a convolution problem: xs = input [N, Ci, H, W], ws = weight [Co, Ci, kh, kw], os_ = output
    [N, Co, Ho, Wo]; direction 'F' (forward), 'B' (data gradient), 'W' (weight gradient); channels_last problems spell the three
    layouts out (`NHWC-NHWC-NHWC`), NCHW ones a single `NCHW`; dtype BF16 / FP32 / FP16."""
    N, Ci, H, W = xs
    Co, _, kh, kw = ws
    Ho, Wo = os_[2], os_[3]
    a = (Ci, H, W, Co, Ho, Wo) if direction == 'F' else (Co, Ho, Wo, Ci, H, W)
    return '%d-%d-%d-%dx%d-%d-%d-%d-%d-%dx%d-%dx%d-1x1-0-%s-%s-%s' % (
        a[0], a[1], a[2], kh, kw, a[3], a[4], a[5], N, padding, padding, stride, stride,
        'NHWC-NHWC-NHWC' if nhwc else 'NCHW', dtype, direction)


_FDB_DTYPES = {torch.bfloat16: 'BF16', torch.float32: 'FP32', torch.float16: 'FP16'}


def _fdb_covered(direction, x, w, out_shape, stride, padding):
    dt = _FDB_DTYPES.get(x.dtype)
    if not _FDB['on'] or dt is None or w.dtype != x.dtype:
        return False
    nhwc = x.is_contiguous(memory_format=torch.channels_last) and not (x.shape[1] > 1 and x.is_contiguous())
    if not nhwc and not x.is_contiguous():
        return False
    k = (direction, x.shape, w.shape, stride, padding, nhwc, dt)   # layout and dtype are part of the problem
    hit = _FDB['known'].get(k)
    if hit is None:
        hit = fdb_key(direction, tuple(x.shape), tuple(w.shape), tuple(out_shape), stride, padding, nhwc, dt) in _fdb_keys()
        _FDB['known'][k] = hit
        _FDB['hits' if hit else 'misses'] += 1
    return hit


_MIOPEN_MODE_LOCK = _threading.RLock()


def _miopen(covered, fn, *args):
    """fn(*args) in MIOpen immediate mode when the find-db covers the problem, else as the process is configured."""
    if not covered or not torch.backends.cudnn.benchmark:
        return fn(*args)
    # the benchmark flag is process-global: flipping it is safe only while no other thread issues a convolution.  Convolutions are
    # issued by the thread that runs the step (forward, and -- runtime.backward_here -- backward); the lock makes a second issuing
    # thread (autograd's worker with CFL_AUTOGRAD_THREAD=1, a user's own thread) wait instead of seeing the flag half-flipped.
    with _MIOPEN_MODE_LOCK:
        torch._C._set_cudnn_benchmark(False)
        try:
            return fn(*args)
        finally:
            torch._C._set_cudnn_benchmark(True)


# Round 6: the 3 x 3 / stride 1 weight gradients of the bottlenecks (56 x 56 x 64, 28 x 28 x 128, 14 x 14 x 256, 7 x 7 x 512: 30 of the 33
# conv2 layers of a ResNet-101; the three stride-2 ones stay on the library) on the hand-written kernel of csrc/wgrad3x3.hip instead of MIOpen's igemm_wrw -- it moves the operands once and runs at several times
# the library kernel's MFMA rate, and what the side stream does not ask of HBM and of the matrix pipes the main stream gets
# (profiles/r6_bound_wgrad.json: the weight gradients cost the step 6.5 ms).  CFL_NO_WGRAD3=1 / tools/ab_step.py --knob wgrad3 is the A/B.
WGRAD3 = [_os.environ.get('CFL_NO_WGRAD3', '0') != '1']
WGRAD3_TAKEN = [0]


def conv3x3_wgrad(dy, x, weight):
    """dW of y = conv2d(x, weight, stride 1, padding 1) for a 3 x 3 weight, all bf16 channels_last: the kernel of csrc/wgrad3x3.hip.
    None when the shape / layout is not taken (the caller goes to the library)."""
    if not (weight.dim() == 4 and weight.shape[2] == 3 and weight.shape[3] == 3 and x.dim() == 4 and dy.dim() == 4):
        return None
    if not (dy.dtype == x.dtype == weight.dtype == torch.bfloat16 and x.is_cuda):
        return None
    N, Ci, H, W = x.shape
    Co = weight.shape[0]
    if tuple(dy.shape) != (N, Co, H, W) or weight.shape[1] != Ci:
        return None
    lib = _lib.load()
    if not lib.cfl_conv3x3_wgrad_supported(N, H, W, Ci, Co):
        return None
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and dy.is_contiguous(memory_format=cl) and weight.is_contiguous(memory_format=cl)):
        return None
    dw = torch.empty_like(weight)                                        # [Co][3][3][Ci] in memory, as the weight
    ws = _ws(lib.cfl_conv3x3_wgrad_ws_bytes(N, H, W, Ci, Co), x.device)
    _lib.check(lib.cfl_conv3x3_wgrad(_ptr(dy), _ptr(x), N, H, W, Ci, Co, _ptr(dw), _ptr(ws), _stream(x)), 'cfl_conv3x3_wgrad')
    WGRAD3_TAKEN[0] += 1
    return dw


# ... and the 1 x 1 / stride 1 ones (66 per step: conv1 / conv3 of every bottleneck, the stride-1 downsample) on csrc/wgrad1x1.hip instead of
# the library's batched GEMM with atomics + zero fill + cast.  CFL_NO_WGRAD1=1 / tools/ab_step.py --knob wgrad1 is the A/B.
WGRAD1 = [_os.environ.get('CFL_NO_WGRAD1', '0') != '1']
WGRAD1_TAKEN = [0]
# maps up to this height only (0 = all).  Stand-alone the kernel beats the library at 7 x 7 / 14 x 14 (60 + 9 vs 64 us while moving a
# third of its bytes), ties at 28 x 28 and loses at 56 x 56 (163 + 31 vs 114 us: 512 MB of operands against a 16 K-element output, where
# its split-K partials and 128 workgroups are the wrong shape).  In the step (tools/ab_step.py --knob w1hwN, profiles/r6_ab_w1hw*.json):
# all maps 42.81, up to 14: 42.56, up to 28: 42.43 ms -- layer1's nine 1 x 1 weight gradients stay on the library.
WGRAD1_MAX_HW = [int(_os.environ.get('CFL_WGRAD1_MAX_HW', '28'))]


def conv1x1_wgrad(dy, x, weight):
    """dW of y = conv2d(x, weight) for a 1 x 1 / stride 1 weight, all bf16 channels_last: the kernel of csrc/wgrad1x1.hip.  None when
    the shape / layout is not taken (the caller goes to the library)."""
    if not (weight.dim() == 4 and weight.shape[2] == 1 and weight.shape[3] == 1 and x.dim() == 4 and dy.dim() == 4):
        return None
    if not (dy.dtype == x.dtype == weight.dtype == torch.bfloat16 and x.is_cuda):
        return None
    N, Ci, H, W = x.shape
    Co = weight.shape[0]
    if tuple(dy.shape) != (N, Co, H, W) or weight.shape[1] != Ci:
        return None
    M = N * H * W
    if WGRAD1_MAX_HW[0] and H > WGRAD1_MAX_HW[0]:
        return None
    lib = _lib.load()
    if not lib.cfl_conv1x1_wgrad_supported(M, Ci, Co):
        return None
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and dy.is_contiguous(memory_format=cl)):
        return None
    if not (weight.is_contiguous() or weight.is_contiguous(memory_format=cl)):
        return None
    dw = torch.empty_like(weight)                                        # [Co][Ci] in memory in either format
    ws = _ws(lib.cfl_conv1x1_wgrad_ws_bytes(M, Ci, Co), x.device)
    _lib.check(lib.cfl_conv1x1_wgrad(_ptr(dy), _ptr(x), M, Ci, Co, _ptr(dw), _ptr(ws), _stream(x)), 'cfl_conv1x1_wgrad')
    WGRAD1_TAKEN[0] += 1
    return dw


def _conv_wgrad(args):
    dy, x, w = args[0], args[1], args[2]
    if WGRAD3[0] and w.shape[2] == 3 and args[4][0] == 1 and args[5][0] == 1:
        g = conv3x3_wgrad(dy, x, w)
        if g is not None:
            return g
    if WGRAD1[0] and w.shape[2] == 1 and w.shape[3] == 1 and args[4][0] == 1 and args[5][0] == 0:
        g = conv1x1_wgrad(dy, x, w)
        if g is not None:
            return g
    if X3CONV[0] and X3WGRAD[0] and w.dtype == torch.float32 and w.shape[2] == 3 and w.shape[3] == 3 and args[4][0] == 1 and args[5][0] == 1:
        g = conv3x3_x3_wgrad(dy, x, w)
        if g is not None:
            return g
    cov = _fdb_covered('W', x, w, dy.shape, args[4][0], args[5][0])
    return _miopen(cov, torch.ops.aten.convolution_backward, *args, [False, True, False])[1]


# Round 6: the 3 x 3 / stride 1 / padding 1 convolutions of fp32 channels_last tensors (the clients' ResNet-18 BasicBlocks: 16 of its
# 20 convolutions) on csrc/conv3x3_x3.hip -- fp32-class accuracy on the bf16 matrix pipe (3 x bf16 split), forward and data gradient;
# the weight gradient: conv3x3_x3_wgrad below.  `--client_conv_x3` (creamfl_amd/flags.py) / CFL_X3CONV=1 switch it on.
X3CONV = [_os.environ.get('CFL_X3CONV', '0') == '1']
X3CONV_TAKEN = [0]
X3CONV_S2 = [_os.environ.get('CFL_NO_X3CONV_S2', '0') != '1']       # the stride-2 forward form (A/B: CFL_NO_X3CONV_S2=1 -> library)


def conv3x3_x3_supported(x, w, stride, padding):
    if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 4 and w.dim() == 4):
        return False
    if not (w.shape[2] == 3 and w.shape[3] == 3 and stride in (1, 2) and padding == 1 and w.shape[1] == x.shape[1]):
        return False
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and w.is_contiguous(memory_format=cl)):
        return False
    N, Ci, H, W = x.shape
    if stride == 2 and (not X3CONV_S2[0] or H % 2 or W % 2 or W > 62):     # (forward only: the down-sampling convolutions)
        return False
    return bool(_lib.load().cfl_conv3x3_x3_supported(N, H, W, Ci, w.shape[0]))


# weight images of FROZEN weights (requires_grad False: the clients' old model, evaluation) are kept across calls.  An entry belongs
# to the tensor OBJECT (a weak reference whose death removes it: a recycled address or id cannot revive it) and holds the storage
# address and version counter it was built from.  Trainable weights are never cached: inside a HIP-graph capture a cache hit would
# leave the image's build out of the graph while the replays train the weight.
_X3_IMAGES = {}


def _x3_weight_image(w, rot=False):
    import weakref
    lib = _lib.load()
    Co, Ci = w.shape[0], w.shape[1]
    frozen = not w.requires_grad and w.grad_fn is None
    key = (id(w), bool(rot))
    if frozen:
        hit = _X3_IMAGES.get(key)
        if hit is not None and hit[0]() is w and hit[1] == (w.data_ptr(), w._version, Co, Ci):
            return hit[2]
    img = _ws(lib.cfl_conv3x3_x3_wimage_bytes(Ci, Co), w.device)
    if rot:
        _lib.check(lib.cfl_conv3x3_x3_wimage_rot(_ptr(w), Ci, Co, _ptr(img), _stream(w)), 'cfl_conv3x3_x3_wimage_rot')
    else:
        _lib.check(lib.cfl_conv3x3_x3_wimage(_ptr(w), Ci, Co, _ptr(img), _stream(w)), 'cfl_conv3x3_x3_wimage')
    if frozen and not torch.cuda.is_current_stream_capturing():      # (a buffer of a graph's private pool must not outlive the graph)
        _X3_IMAGES[key] = (weakref.ref(w, lambda _r, k=key: _X3_IMAGES.pop(k, None)), (w.data_ptr(), w._version, Co, Ci), img)
    return img


def conv3x3_x3_forward(x, w, variant=0, rotated=False, stride=1):
    """conv2d(x, w, stride 1, padding 1) for fp32 channels_last x [N, Ci, H, W] and w [Co, Ci, 3, 3] (csrc/conv3x3_x3.hip); no
    autograd (the Functions that own the convolutions call it for their forward and, on the rotated weight, their data gradient).
    variant 0 / >= 200: version 3 (the weight split once into an image of the kernel's LDS stage; maps up to 63 wide); 21 .. 142: the
    earlier kernels (any width).  rotated=True: w is the FORWARD weight [Ci_out_of_this_call ... ] of the convolution whose data
    gradient this is -- x is dY [N, Co_w, H, W], the result dX [N, Ci_w, H, W] (the image of the rotated weight is built directly)."""
    lib = _lib.load()
    N, Cin, H, W = x.shape
    if rotated:
        assert w.shape[0] == Cin
        Cout = w.shape[1]
    else:
        Cout = w.shape[0]
    if stride == 2:                                        # forward of a down-sampling convolution (even H, W; the stride-1 weight image)
        assert not rotated and H % 2 == 0 and W % 2 == 0
        y = torch.empty((N, Cout, H // 2, W // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        img = _x3_weight_image(w)
        _lib.check(lib.cfl_conv3x3_x3_fwd_img_s2(_ptr(x), _ptr(img), N, H, W, Cin, Cout, _ptr(y), _stream(x)), 'cfl_conv3x3_x3_fwd_img_s2')
        X3CONV_TAKEN[0] += 1
        return y
    y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if (variant == 0 or variant >= 200) and W <= 63:
        img = _x3_weight_image(w, rot=rotated)
        _lib.check(lib.cfl_conv3x3_x3_fwd_img(_ptr(x), _ptr(img), N, H, W, Cin, Cout, _ptr(y), int(variant), _stream(x)), 'cfl_conv3x3_x3_fwd_img')
    else:
        wk = conv3x3_x3_rotated(w) if rotated else w
        _lib.check(lib.cfl_conv3x3_x3_fwd(_ptr(x), _ptr(wk), N, H, W, Cin, Cout, _ptr(y), int(variant), _stream(x)), 'cfl_conv3x3_x3_fwd')
    X3CONV_TAKEN[0] += 1
    return y


def conv3x3_x3_rotated(w):
    """W'[ci][co][kh][kw] = W[co][ci][2 - kh][2 - kw] as a channels_last [Ci, Co, 3, 3] tensor: conv(dY, W') is the data gradient."""
    Co, Ci = w.shape[0], w.shape[1]
    wr = torch.empty((Ci, Co, 3, 3), dtype=torch.float32, device=w.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().cfl_conv3x3_x3_rot_weight(_ptr(w), Ci, Co, _ptr(wr), _stream(w)), 'cfl_conv3x3_x3_rot_weight')
    return wr


# ... and their WEIGHT gradient on csrc/wgrad3x3_x3.hip (H = W in {7, 14, 28, 56}, channel counts multiples of 64: the thirteen stride-1
# BasicBlock convolutions of ResNet-18).  CFL_NO_X3WGRAD=1 leaves it on the library while the forward / data gradient stay here.
X3WGRAD = [_os.environ.get('CFL_NO_X3WGRAD', '0') != '1']
X3WGRAD_TAKEN = [0]


def conv3x3_x3_wgrad(dy, x, weight):
    """dW of y = conv2d(x, weight, stride 1, padding 1) for a 3 x 3 weight, all fp32 channels_last, products as 3 x bf16-split MFMAs
    (csrc/wgrad3x3_x3.hip).  None when the shape / layout is not taken (the caller goes to the library)."""
    if not (weight.dim() == 4 and weight.shape[2] == 3 and weight.shape[3] == 3 and x.dim() == 4 and dy.dim() == 4):
        return None
    if not (dy.dtype == x.dtype == weight.dtype == torch.float32 and x.is_cuda):
        return None
    N, Ci, H, W = x.shape
    Co = weight.shape[0]
    if tuple(dy.shape) != (N, Co, H, W) or weight.shape[1] != Ci:
        return None
    lib = _lib.load()
    if not lib.cfl_conv3x3_x3_wgrad_supported(N, H, W, Ci, Co):
        return None
    cl = torch.channels_last
    if not (x.is_contiguous(memory_format=cl) and dy.is_contiguous(memory_format=cl) and weight.is_contiguous(memory_format=cl)):
        return None
    dw = torch.empty_like(weight)                                        # [Co][3][3][Ci] in memory, as the weight
    ws = _ws(lib.cfl_conv3x3_x3_wgrad_ws_bytes(N, H, W, Ci, Co), x.device)
    _lib.check(lib.cfl_conv3x3_x3_wgrad(_ptr(dy), _ptr(x), N, H, W, Ci, Co, _ptr(dw), _ptr(ws), _stream(x)), 'cfl_conv3x3_x3_wgrad')
    X3WGRAD_TAKEN[0] += 1
    return dw


def _conv_dgrad(args):
    dy, x, w = args[0], args[1], args[2]
    if X3CONV[0] and w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3 and args[4][0] == 1 and args[5][0] == 1 \
            and dy.is_cuda and dy.dtype == torch.float32 and w.dtype == torch.float32 and dy.dim() == 4 and dy.shape[1] == w.shape[0] \
            and dy.is_contiguous(memory_format=torch.channels_last) and w.is_contiguous(memory_format=torch.channels_last) \
            and _lib.load().cfl_conv3x3_x3_supported(dy.shape[0], dy.shape[2], dy.shape[3], w.shape[0], w.shape[1]):
        # dX = conv(dY, W') on the rotated, transposed weight: the forward kernel with the roles of the channel counts swapped
        return conv3x3_x3_forward(dy, w, rotated=True)
    cov = _fdb_covered('B', x, w, dy.shape, args[4][0], args[5][0])
    return _miopen(cov, torch.ops.aten.convolution_backward, *args, [True, False, False])[0]


def _conv_fwd(x, w, stride, padding):
    if X3CONV[0] and conv3x3_x3_supported(x, w, stride, padding):
        return conv3x3_x3_forward(x, w, stride=stride)
    kh = w.shape[2]
    out_shape = (x.shape[0], w.shape[0], (x.shape[2] + 2 * padding - kh) // stride + 1, (x.shape[3] + 2 * padding - w.shape[3]) // stride + 1)
    return _miopen(_fdb_covered('F', x, w, out_shape, stride, padding), torch.nn.functional.conv2d, x, w, None, stride, padding)


class _ConvSplitFn(torch.autograd.Function):
    """y = conv2d(x, w, stride, padding) (no bias, groups 1) with the backward split in two:
      * the DATA gradient stays on the critical path (main stream); for 1x1 / stride-1 kernels it runs on the
        hand-written bf16 MFMA GEMM (dX[M, Ci] = dY[M, Co] @ W[Co, Ci]), which beats MIOpen's backward-data kernels on
        every ResNet-101 shape (docs/history/tools/wgrad_probe.py vs tools/kernel_bench.py --cases gemm16);
      * the WEIGHT gradient, which nothing but the optimizer waits for, is issued on the auxiliary 'wgrad' stream and
        overlaps the HBM-bound BN / data-gradient kernels of the layers below (streams.py).  MIOpen computes it."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, gemm_dgrad, side_wgrad, stats_nblk=0):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, gemm_dgrad, side_wgrad)
        # weight gradient inside the BatchNorm backward that follows (WGRAD_FUSE below): a per-call holder both nodes share
        ctx.wg = None
        if (WGRAD_FUSE[0] and gemm_dgrad and side_wgrad and ctx.needs_input_grad[1] and weight.dtype == torch.bfloat16
                and _lib.load().cfl_bn_bwd_wgrad_supported(x.shape[0] * x.shape[2] * x.shape[3], weight.shape[0], weight.shape[1])):
            ctx.wg = _LAST_WG[0] = {'x': x, 'w': weight, 'done': False}
        tok = getattr(x, '_cfl_tok', 0)
        ctx.x_tok = tok if (gemm_dgrad and ctx.needs_input_grad[0] and tok and tok in JOIN['mask']) else 0
        if ctx.x_tok:
            JOIN['consumer'].add(tok)                    # this node's data gradient can take the gradient join of x (JOIN above)
        if stats_nblk:
            # a training-mode BatchNorm follows: the forward GEMM on the B-resident streaming kernel with that layer's batch
            # statistics in its epilogue (csrc/gemm_bf16.hip: cfl_gemm_bf16_nt_stats); conv_split hands the partials on
            N, Ci, H, W = x.shape
            Co = weight.shape[0]
            y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            pstat = torch.empty(2 * stats_nblk * Co, dtype=torch.float32, device=x.device)
            GEMM_COUNTERS['flops'] += 2 * N * H * W * Co * Ci
            GEMM_COUNTERS['bytes'] += 2 * (N * H * W * (Ci + Co) + Co * Ci) + 8 * stats_nblk * Co
            _lib.check(_lib.load().cfl_gemm_bf16_nt_stats(_ptr(x), Ci, _ptr(weight), Ci, _ptr(y), N * H * W, Co, Ci, _ptr(pstat),
                                                          _stream(x)), 'cfl_gemm_bf16_nt_stats')
            _LAST_STATS[0] = (pstat, stats_nblk, N * H * W, Co)
            return y
        return _conv_fwd(x, weight, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding, gemm_dgrad, side_wgrad = ctx.cfg
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        nhwc = x.is_contiguous(memory_format=torch.channels_last) and not (x.shape[1] > 1 and x.is_contiguous())
        dy = dy.contiguous(memory_format=torch.channels_last) if nhwc else dy.contiguous()
        args = (dy, x, weight, None, [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1)
        dx = dw = None
        if ctx.wg is not None:
            ctx.wg['x'] = None                                        # (the holder must not outlive the step's activations)
        if ctx.needs_input_grad[1] and not (ctx.wg is not None and ctx.wg['done']):
            if side_wgrad:
                from . import streams
                if streams.DEFER_WGRAD[0]:
                    # Autograd gets no weight gradient from this node (None = zero); when the queue is flushed the
                    # gradient is computed on the 'wgrad' stream and accumulated into `weight.grad` directly, exactly
                    # what AccumulateGrad would have done (a tensor handed over now and filled later does not work:
                    # AccumulateGrad clones a gradient that something else still references).
                    def task(main, side, args=args, weight=weight):
                        g = _conv_wgrad(args)
                        args[0].record_stream(side)
                        args[1].record_stream(side)
                        g.record_stream(main)
                        with torch.no_grad():
                            if weight.grad is None:
                                weight.grad = g
                            else:
                                weight.grad.add_(g)
                        if streams.GRAD_READY[0] is not None:
                            streams.GRAD_READY[0](weight)              # multi-GPU: this gradient may now be bucketed
                    streams.defer(task)
                else:
                    main = torch.cuda.current_stream(x.device)
                    side = streams.get(x.device, 'wgrad')
                    side.wait_stream(main)                               # dy (and x) are ready on the main stream
                    with torch.cuda.stream(side):
                        dw = _conv_wgrad(args)
                    dy.record_stream(side)
                    x.record_stream(side)
                    dw.record_stream(main)
                _queue_stream_join(x.device)
            else:
                dw = _conv_wgrad(args)
        if ctx.needs_input_grad[0]:
            if gemm_dgrad:
                dx = torch.empty_like(x)                                 # channels_last: the [M, Ci] matrix
                wt = _WT['views'].get(weight.data_ptr()) if _WT['valid'] else None     # prepared in one launch?
                if wt is None:
                    wt = torch.empty(Ci, Co, dtype=torch.bfloat16, device=x.device)   # [Ci, Co]: reduction axis contiguous
                    _lib.check(_lib.load().cfl_transpose_bf16(_ptr(weight), Co, Ci, _ptr(wt), _stream(x)), 'cfl_transpose_bf16')
                skip = JOIN['pending'].pop(ctx.x_tok, None) if (JOIN['on'] and ctx.x_tok) else None
                if skip is not None:
                    # dX = (dY W + skip gradient) . ReLU mask of the BatchNorm that produced x: pre-joined for that layer
                    gemm_bf16_nt(dy.permute(0, 2, 3, 1).reshape(N * H * W, Co), wt, out=dx.permute(0, 2, 3, 1).reshape(N * H * W, Ci),
                                 add=skip, mask=JOIN['mask'][ctx.x_tok])
                    # the BatchNorm backward must be handed exactly this tensor.  The registry keeps a REFERENCE to it next to
                    # the address: were there a second consumer of the pre-joined output, autograd's input buffer could
                    # otherwise add the other (unmasked) gradient into this freshly allocated dx IN PLACE (it does so when it
                    # holds the only reference) and the sum would keep dx's address; with the reference held it must allocate
                    # a new tensor, and the address check below sees it whatever the arrival order
                    JOIN['pre'][ctx.x_tok] = (dx.data_ptr(), dx)
                    JOIN['fused'] += 1
                elif DGRAD_PLAIN_LIB[0] and Co >= DGRAD_PLAIN_LIB[0]:
                    # measurement knob (tools/ab_step.py --knob dgradlib): the un-joined data gradient as the library's FORWARD
                    # 1x1 convolution on the prepared W^T
                    dx = torch.nn.functional.conv2d(dy, wt.view(Ci, Co, 1, 1))
                else:
                    gemm_bf16_nt(dy.permute(0, 2, 3, 1).reshape(N * H * W, Co), wt, out=dx.permute(0, 2, 3, 1).reshape(N * H * W, Ci))
            elif (nhwc and stride == 1 and padding == weight.shape[2] // 2 and weight.dtype == torch.bfloat16 and _rot_ok(weight)
                  and not _NO_FWD_DGRAD):
                # k x k / stride 1 / same padding: dX = conv2d(dY, W') with the rotated, transposed weight -- MIOpen's FORWARD
                # kernels run this 1.3-1.6x faster than its backward-data kernels on every ResNet-101 shape (3x3, batch 256:
                # 56x56x64 169 -> 134 us, 28x28x128 126 -> 97, 14x14x256 120 -> 75, 7x7x512 138 -> 104; docs/history/tools/dgrad3x3_probe.py)
                wr = _WT['views'].get(weight.data_ptr()) if _WT['valid'] else None
                if wr is None and dy.numel() >= (1 << 22):
                    # not prepared (client trainers): two small kernels, repaid by the faster convolution on large maps only
                    wr = weight.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
                if wr is not None:
                    dx = _conv_fwd(dy, wr, 1, padding)
                else:
                    dx = _conv_dgrad(args)
            else:
                dx = _conv_dgrad(args)
        return dx, dw, None, None, None, None, None


# The weight gradient of a bottleneck's conv3 inside the BatchNorm backward-apply pass that produces its dY (csrc/bnorm.hip:
# cfl_bn_bwd_wgrad; round 5).  The convolution's forward leaves a holder {x: its input, w: its weight, done} on its output tensor
# object; bn_act_train hands it to the BatchNorm node; in the backward, when the BatchNorm's gradient arrives pre-joined (ops.JOIN)
# and the shape is taken, ONE launch sequence writes dY, dgamma, dbeta AND dW, accumulates dW into weight.grad (as the deferred
# weight gradients do) and marks the holder done -- the convolution's own backward then skips its library weight gradient.
# MEASURED AND NOT SHIPPED (off by default, CFL_WGRAD_FUSE=1 / tools/ab_step.py --knob wgfuse switch it on): stand-alone the fused
# pass costs +19 us per layer3 block for a weight gradient the library takes 63 us for (tools/hip/bnwg_probe.hip), but inside the
# server step it LOSES: 44.47 vs 43.93 ms (six alternating rounds, profiles/r5_ab_wgfuse.json) -- the BatchNorm apply pass is on
# the main stream, i.e. on the step's critical path, and every microsecond added there shows, while the library kernel it replaces
# runs on the side stream, which only costs the step through contention (the data-gradient GEMMs did speed up, 100 -> 88 us).
WGRAD_FUSE = [_os.environ.get('CFL_WGRAD_FUSE', '0') == '1']
WGRAD_FUSED = [0]                                                   # launches taken (tests / benches read it)
_LAST_WG = [None]
CONV_STATS = [_os.environ.get('CFL_NO_CONV_STATS', '0') != '1']     # switch (also flipped by tools/ab_step.py --knob convstats)
CONV_STATS_MIN_M = 32768        # below: too few 32-row tiles per wave for the streaming kernel (as for the joined data gradient)
_LAST_STATS = [None]


def conv_split(x, weight, stride=1, padding=0, side_wgrad=True, bn_follows=False):
    """Trunk convolution with the split backward of _ConvSplitFn (x: channels_last HIP tensor; weight [Co, Ci, k, k]).
    bn_follows: a training-mode BatchNorm consumes the output -- where the shape allows, the forward runs on the hand-written
    streaming GEMM with that BatchNorm's statistics in its epilogue, and the output tensor object carries the partial sums
    (`_cfl_bnstats`) for bn_act_train to pick up (no statistics pass over the 4-planes-wide tensor)."""
    gemm = (weight.shape[2] == 1 and weight.shape[3] == 1 and stride == 1 and padding == 0 and conv1x1_supported(x, weight))
    nblk = 0
    if bn_follows and gemm and CONV_STATS[0] and x.shape[0] * x.shape[2] * x.shape[3] >= CONV_STATS_MIN_M:
        nblk = int(_lib.load().cfl_gemm_bf16_nt_stats_nblk(x.shape[0] * x.shape[2] * x.shape[3], weight.shape[0], weight.shape[1]))
    _LAST_WG[0] = None
    y = _ConvSplitFn.apply(x, weight, int(stride), int(padding), bool(gemm), bool(side_wgrad), nblk)
    if nblk:
        y._cfl_bnstats, _LAST_STATS[0] = _LAST_STATS[0], None
    if _LAST_WG[0] is not None:
        if bn_follows:
            y._cfl_wg = _LAST_WG[0]                       # bn_act_train picks it up; anything in between drops the fusion
        _LAST_WG[0] = None
    return y


def conv_gated(x, weight, stride=1, padding=0):
    """A plain convolution (any layout / dtype the library takes; no bias, groups 1) whose three library calls -- forward, data
    gradient, weight gradient -- each answer from the shipped find-db where it holds the problem (MIOpen immediate mode for that
    call only) instead of PyTorch's timed search: the client encoders' fp32 NCHW convolutions and every other convolution outside
    the channels_last bf16 trunk path (round 5; the first round of a configs[2] federation spent 704 s in those searches)."""
    return _ConvSplitFn.apply(x, weight, int(stride), int(padding), False, False, 0)


def conv_gate_worthwhile(x, weight, stride, padding):
    """True when at least the FORWARD problem of this convolution is in the shipped find-db (else the plain library call)."""
    if not (_FDB['on'] and x.is_cuda and x.dim() == 4 and weight.dim() == 4):
        return False
    out_shape = (x.shape[0], weight.shape[0], (x.shape[2] + 2 * padding - weight.shape[2]) // stride + 1,
                 (x.shape[3] + 2 * padding - weight.shape[3]) // stride + 1)
    return _fdb_covered('F', x, weight, out_shape, stride, padding)


def conv1x1(x, weight):
    return conv_split(x, weight, 1, 0)


# ---- ResNet stem: 7x7 / stride 2 / pad 3 on 3 channels as a 4x4 / stride-1 convolution of the space-to-depth image -------------
def _stem_weight_s2d(w):
    """[Co, 3, 7, 7] -> [Co, 16, 4, 4] (channels_last): W4[co, (2p+q)*3 + c, a+2, b+2] = W[co, c, 2a+p+3, 2b+q+3], a, b in -2..1,
    zero where the 7x7 tap does not exist (2a+p+3 = -1) and in the 4 padding channels."""
    co = w.shape[0]
    wp = torch.nn.functional.pad(w, (1, 0, 1, 0))                                     # tap index kh + 1 = 2 (a + 2) + p
    v = wp.reshape(co, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(co, 12, 4, 4)
    return torch.nn.functional.pad(v, (0, 0, 0, 0, 0, 4)).contiguous(memory_format=torch.channels_last)


def _stem_weight_s2d_inverse(g4, like):
    """gradient of _stem_weight_s2d: [Co, 16, 4, 4] -> [Co, 3, 7, 7] in `like`'s memory format."""
    co = g4.shape[0]
    g = g4[:, :12].reshape(co, 2, 2, 3, 4, 4).permute(0, 3, 4, 1, 5, 2).reshape(co, 3, 8, 8)[:, :, 1:, 1:]
    return g.contiguous(memory_format=torch.channels_last if like.is_contiguous(memory_format=torch.channels_last)
                        and not like.is_contiguous() else torch.contiguous_format)


class _StemConvFn(torch.autograd.Function):
    """torchvision ResNet.conv1 (image_encoder.py:27-36).  MIOpen runs the problem as written at 376 us forward / 406 us weight
    gradient (batch 256: 3-channel, 6-byte pixels, K = 147); after space-to-depth (csrc/pool.hip: cfl_stem_s2d, one pass, 16
    channels) the same convolution is a 4x4 / stride-1 one that its kernels run in 229 / 225 us (docs/history/tools/stem_probe.py).  The input
    needs no gradient (images); the weight gradient is deferred to the auxiliary stream like every trunk convolution's."""

    @staticmethod
    def forward(ctx, x, weight, side_wgrad):
        lib = _lib.load()
        N, _, H, W = x.shape
        xs = torch.empty((N, 16, H // 2 + 3, W // 2 + 3), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        _lib.check(lib.cfl_stem_s2d(_ptr(x), int(x.dtype == torch.float32), N, H, W, _ptr(xs), _stream(x)), 'cfl_stem_s2d')
        w4 = _stem_weight_s2d(weight.detach())
        ctx.save_for_backward(xs, w4, weight)
        ctx.side_wgrad = side_wgrad
        with torch.autocast('cuda', enabled=False):
            return _conv_fwd(xs, w4, 1, 0)

    @staticmethod
    def backward(ctx, dy):
        xs, w4, weight = ctx.saved_tensors
        if dy.dtype != xs.dtype:
            dy = dy.to(xs.dtype)
        dy = dy.contiguous(memory_format=torch.channels_last)
        args = (dy, xs, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        if not ctx.needs_input_grad[1]:
            return None, None, None

        def wgrad():
            return _stem_weight_s2d_inverse(_conv_wgrad(args), weight)
        from . import streams
        if ctx.side_wgrad and streams.DEFER_WGRAD[0]:
            def task(main, side, weight=weight):
                g = wgrad()
                args[0].record_stream(side)
                args[1].record_stream(side)
                g.record_stream(main)
                with torch.no_grad():
                    if weight.grad is None:
                        weight.grad = g
                    else:
                        weight.grad.add_(g)
                if streams.GRAD_READY[0] is not None:
                    streams.GRAD_READY[0](weight)
            streams.defer(task)
            _queue_stream_join(xs.device)
            return None, None, None
        return None, wgrad(), None


def stem_conv_supported(x, weight, stride, padding):
    """bf16 weights; bf16 images, or fp32 images under bf16 autocast (the cast autocast would do is folded into the kernel)."""
    xok = x.dtype == torch.bfloat16 or (x.dtype == torch.float32 and torch.is_autocast_enabled('cuda')
                                        and torch.get_autocast_dtype('cuda') == torch.bfloat16)
    return (x.is_cuda and x.dim() == 4 and xok and weight.dtype == torch.bfloat16 and not x.requires_grad
            and tuple(weight.shape[1:]) == (3, 7, 7) and stride == 2 and padding == 3 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and not _NO_STEM_S2D)


def stem_conv(x, weight, side_wgrad=True):
    """y = conv2d(x, weight, stride 2, padding 3) for the 3-channel 7x7 ResNet stem (see _StemConvFn)."""
    return _StemConvFn.apply(x, weight, bool(side_wgrad))


# --------------------------------------------------------------------------- ResNet stem max pooling (csrc/pool.hip)
class _MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty(N * Ho * Wo * C, dtype=torch.uint8, device=x.device)
        _lib.check(lib.cfl_maxpool3s2_fwd(_ptr(x), N, H, W, C, _ptr(y), _ptr(idx), _stream(x)), 'cfl_maxpool3s2_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        _lib.check(lib.cfl_maxpool3s2_bwd(_ptr(dy), _ptr(idx), N, H, W, C, _ptr(dx), _stream(dy)), 'cfl_maxpool3s2_bwd')
        return dx


def maxpool3s2(x):
    """3x3 / stride 2 / pad 1 max pooling of a channels_last bf16 activation (the ResNet stem pool)."""
    if x.dtype != torch.bfloat16 or not bn_act_supported(x, x.shape[1]):
        raise _lib.CreamflHipError('maxpool3s2: expected a channels_last bf16 HIP tensor with C % 8 == 0')
    return _MaxPool3s2Fn.apply(x)


# --------------------------------------------------------------------------- BERT tower glue (csrc/bertfuse.hip)
_DROPOUT_CALLS = [0]


def _next_dropout_seed():
    """Deterministic per call: a counter mixed with torch's seed (torch.manual_seed reproduces the masks)."""
    _DROPOUT_CALLS[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _DROPOUT_CALLS[0] * 0x85EBCA6B) & 0xFFFFFFFF


_DROPOUT_TICK = [None]


def dropout_tick(device):
    """The device word that varies the fused dropout masks between replays of a captured step (cfl_set_dropout_tick): created and
    registered with the library on first use -- from then on every fused dropout launch of the process adds it to its seed, eager
    steps included.  A step that is (or will be) captured calls `dropout_tick(device).add_(1)` once at its start."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if _DROPOUT_TICK[0] is None:
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.load().cfl_set_dropout_tick(t.data_ptr()), 'cfl_set_dropout_tick')
        _DROPOUT_TICK[0] = t
    elif _DROPOUT_TICK[0].device != dev:
        # the library holds ONE registered pointer per process (one process drives one GPU: DESIGN section 6); a second device's
        # kernels would dereference the first device's word
        raise _lib.CreamflHipError('dropout_tick: registered on %s, asked for %s -- one process drives one GPU (launch one rank per '
                                   'device); the fused dropout tick is a per-process word' % (_DROPOUT_TICK[0].device, dev))
    return _DROPOUT_TICK[0]


def _bf16c(t, name):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.CreamflHipError(f'{name}: expected a CUDA/HIP tensor (no CPU fallback in creamfl_amd)')
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    return t.contiguous()


def _alias(t):
    """A second tensor object on the same storage (not an autograd view): lets one buffer be returned as two
    outputs of an autograd Function, so that their gradients arrive separately."""
    return t.new_empty(0).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


class _DalnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, bias, residual, gamma, beta, p, eps, seed):
        lib = _lib.load()
        H = g.shape[-1]
        T = g.numel() // H
        need = any(ctx.needs_input_grad[:5])
        z = torch.empty_like(g)
        s = torch.empty_like(g) if need else None
        stats = torch.empty(2, T, dtype=torch.float32, device=g.device) if need else None
        _lib.check(lib.cfl_daln_fwd(_ptr(g), _ptr(bias), int(bias is not None and bias.dtype == torch.bfloat16), _ptr(residual),
                                    _ptr(gamma), _ptr(beta), T, H, eps, p, seed, _ptr(z), _ptr(s),
                                    _ptr(stats), _ptr(stats[1]) if need else None, _stream(g)), 'cfl_daln_fwd')
        if need:
            ctx.save_for_backward(s, gamma, stats)
        ctx.hyper = (p, seed, T, H, None if bias is None else (bias.dtype, bias.shape))
        ctx.set_materialize_grads(False)                  # an unused alias must arrive as None, not as a zero tensor
        return z, _alias(z)

    @staticmethod
    def backward(ctx, dza, dzb):
        lib = _lib.load()
        s, gamma, stats = ctx.saved_tensors
        p, seed, T, H, bias_meta = ctx.hyper
        if dza is None:
            dza, dzb = dzb, None
        if dza is None:
            return (None,) * 8
        dza = _bf16c(dza, 'dz')
        dzb = _bf16c(dzb, 'dz') if dzb is not None else None
        ds = torch.empty_like(s)
        dy = torch.empty_like(s) if p > 0 else None
        dgb = torch.empty(2, H, dtype=torch.float32, device=s.device)
        dbias = torch.empty(bias_meta[1], dtype=bias_meta[0], device=s.device) if (bias_meta and ctx.needs_input_grad[1]) else None
        ws = _ws(lib.cfl_daln_ws_bytes(T, H), s.device)
        _lib.check(lib.cfl_daln_bwd(_ptr(s), _ptr(dza), _ptr(dzb), _ptr(gamma), _ptr(stats), _ptr(stats[1]), T, H, p, seed,
                                    _ptr(ds), _ptr(dy), _ptr(dgb), _ptr(dbias),
                                    int(dbias is not None and dbias.dtype == torch.bfloat16), _ptr(ws), _stream(s)), 'cfl_daln_bwd')
        return (dy if dy is not None else ds), dbias, ds, dgb[0], dgb[1], None, None, None


class _PreLnFn(torch.autograd.Function):
    """(s, z) = (g + bias + residual, LayerNorm(s)): the tail of a pre-LN sub-layer fused with the LayerNorm of the next one."""
    @staticmethod
    def forward(ctx, g, bias, residual, gamma, beta, eps):
        lib = _lib.load()
        H = g.shape[-1]
        T = g.numel() // H
        need = any(ctx.needs_input_grad[:5])
        z = torch.empty_like(g)
        s = torch.empty_like(g)
        stats = torch.empty(2, T, dtype=torch.float32, device=g.device)
        _lib.check(lib.cfl_daln_fwd(_ptr(g), _ptr(bias), int(bias is not None and bias.dtype == torch.bfloat16), _ptr(residual),
                                    _ptr(gamma), _ptr(beta), T, H, eps, 0.0, 0, _ptr(z), _ptr(s), _ptr(stats), _ptr(stats[1]),
                                    _stream(g)), 'cfl_daln_fwd')
        if need:
            ctx.save_for_backward(s, gamma, stats)
        ctx.hyper = (T, H, None if bias is None else (bias.dtype, bias.shape))
        ctx.set_materialize_grads(False)
        return s, z

    @staticmethod
    def backward(ctx, ds_direct, dz):
        lib = _lib.load()
        s, gamma, stats = ctx.saved_tensors
        T, H, bias_meta = ctx.hyper
        if dz is None and ds_direct is None:
            return (None,) * 6
        if dz is None:                                     # (the LayerNorm output unused: only the residual stream carries a gradient)
            dz = torch.zeros_like(s)
        dz = _bf16c(dz, 'dz')
        ds_direct = _bf16c(ds_direct, 'ds') if ds_direct is not None else None
        ds = torch.empty_like(s)
        dgb = torch.empty(2, H, dtype=torch.float32, device=s.device)
        dbias = torch.empty(bias_meta[1], dtype=bias_meta[0], device=s.device) if (bias_meta and ctx.needs_input_grad[1]) else None
        ws = _ws(lib.cfl_daln_ws_bytes(T, H), s.device)
        _lib.check(lib.cfl_preln_bwd(_ptr(s), _ptr(dz), _ptr(ds_direct), _ptr(gamma), _ptr(stats), _ptr(stats[1]), T, H, _ptr(ds),
                                     _ptr(dgb), _ptr(dbias), int(dbias is not None and dbias.dtype == torch.bfloat16), _ptr(ws),
                                     _stream(s)), 'cfl_preln_bwd')
        return ds, dbias, ds, dgb[0], dgb[1], None


def preln_add_layernorm(g, bias, residual, gamma, beta, eps=1e-6):
    """Tail of a pre-LN sub-layer (ViT block: x + out_proj(attn), x + mlp(...)) fused with the LayerNorm that reads the sum
    (csrc/bertfuse.hip): returns (s, z) = (g + bias + residual in bf16, LayerNorm(s)); g is the bias-free GEMM output."""
    g = _bf16c(g, 'g')
    residual = _bf16c(residual, 'residual')
    if residual.shape != g.shape:
        raise RuntimeError(f'shape mismatch {tuple(g.shape)} vs {tuple(residual.shape)}')
    if bias is not None and not bias.is_contiguous():
        bias = bias.contiguous()
    return _PreLnFn.apply(g, bias, residual, _f32(gamma, 'gamma'), _f32(beta, 'beta'), float(eps))


def bert_dropout_add_layernorm(g, bias, residual, gamma, beta, p=0.0, eps=1e-12, seed=None):
    """LayerNorm(dropout(g + bias) + residual) for the BertSelfOutput / BertOutput sub-layers (csrc/bertfuse.hip).
    g: bias-free GEMM output, bf16 [..., H].  Returns TWO tensors on the same buffer: feed the first to the next
    GEMM and use the second as the next residual -- their gradients are summed inside the fused backward."""
    g = _bf16c(g, 'g')
    residual = _bf16c(residual, 'residual')
    if residual.shape != g.shape:
        raise RuntimeError(f'shape mismatch {tuple(g.shape)} vs {tuple(residual.shape)}')
    if bias is not None and not bias.is_contiguous():
        bias = bias.contiguous()
    if seed is None:
        seed = _next_dropout_seed() if p > 0 else 0
    return _DalnFn.apply(g, bias, residual, _f32(gamma, 'gamma'), _f32(beta, 'beta'), float(p), float(eps), int(seed))


class _BiasGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, bias):
        lib = _lib.load()
        I = g.shape[-1]
        T = g.numel() // I
        h = torch.empty_like(g)
        _lib.check(lib.cfl_bias_gelu_fwd(_ptr(g), _ptr(bias), int(bias is not None and bias.dtype == torch.bfloat16), T, I,
                                         _ptr(h), _stream(g)), 'cfl_bias_gelu_fwd')
        ctx.save_for_backward(g, bias)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        g, bias = ctx.saved_tensors
        I = g.shape[-1]
        T = g.numel() // I
        dh = _bf16c(dh, 'dh')
        du = torch.empty_like(g)
        dbias = torch.empty_like(bias) if (bias is not None and ctx.needs_input_grad[1]) else None
        ws = _ws(lib.cfl_bias_gelu_ws_bytes(T, I), g.device)
        _lib.check(lib.cfl_bias_gelu_bwd(_ptr(g), _ptr(bias), int(bias is not None and bias.dtype == torch.bfloat16), _ptr(dh),
                                         T, I, _ptr(du), _ptr(dbias), int(dbias is not None and dbias.dtype == torch.bfloat16),
                                         _ptr(ws), _stream(g)), 'cfl_bias_gelu_bwd')
        return du, dbias


def bert_bias_gelu(g, bias):
    """gelu(g + bias) (exact erf form) for BertIntermediate; the backward also yields the bias gradient."""
    g = _bf16c(g, 'g')
    if bias is not None and not bias.is_contiguous():
        bias = bias.contiguous()
    return _BiasGeluFn.apply(g, bias)


class _AttnSmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, mask, heads):
        lib = _lib.load()
        B, L, H3 = qkv.shape
        H = H3 // 3
        o = torch.empty(B, L, H, dtype=torch.bfloat16, device=qkv.device)
        base = qkv.data_ptr()
        _lib.check(lib.cfl_attn_small_fwd(ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * H), ctypes.c_void_p(base + 4 * H), H3,
                                          L * H3, _ptr(mask), B, L, heads, H // heads, _ptr(o), H, L * H, _stream(qkv)),
                   'cfl_attn_small_fwd')
        ctx.save_for_backward(qkv, mask)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, mask = ctx.saved_tensors
        B, L, H3 = qkv.shape
        H = H3 // 3
        do = _bf16c(do, 'do')
        dqkv = torch.empty_like(qkv)
        base, gbase = qkv.data_ptr(), dqkv.data_ptr()
        _lib.check(lib.cfl_attn_small_bwd(ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * H), ctypes.c_void_p(base + 4 * H), H3,
                                          L * H3, _ptr(mask), B, L, ctx.heads, H // ctx.heads, _ptr(do), H, L * H,
                                          ctypes.c_void_p(gbase), ctypes.c_void_p(gbase + 2 * H), ctypes.c_void_p(gbase + 4 * H),
                                          H3, L * H3, _stream(qkv)), 'cfl_attn_small_bwd')
        return dqkv, None, None


class _AttnSmallVarlenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu, heads):
        lib = _lib.load()
        T, H3 = qkv.shape
        H = H3 // 3
        B = cu.numel() - 1
        o = torch.empty(T, H, dtype=torch.bfloat16, device=qkv.device)
        base = qkv.data_ptr()
        _lib.check(lib.cfl_attn_small_fwd_varlen(ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * H), ctypes.c_void_p(base + 4 * H),
                                                 H3, _ptr(cu), B, heads, H // heads, _ptr(o), H, _stream(qkv)),
                   'cfl_attn_small_fwd_varlen')
        ctx.save_for_backward(qkv, cu)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, cu = ctx.saved_tensors
        T, H3 = qkv.shape
        H = H3 // 3
        do = _bf16c(do, 'do')
        dqkv = torch.empty_like(qkv)              # every row belongs to a sequence: fully written
        base, gbase = qkv.data_ptr(), dqkv.data_ptr()
        _lib.check(lib.cfl_attn_small_bwd_varlen(ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * H), ctypes.c_void_p(base + 4 * H),
                                                 H3, _ptr(cu), cu.numel() - 1, ctx.heads, H // ctx.heads, _ptr(do), H,
                                                 ctypes.c_void_p(gbase), ctypes.c_void_p(gbase + 2 * H),
                                                 ctypes.c_void_p(gbase + 4 * H), H3, _stream(qkv)), 'cfl_attn_small_bwd_varlen')
        return dqkv, None, None


def bert_attention_varlen(qkv, cu_seqlens, heads):
    """The same attention on PACKED tokens: qkv [T, 3H] (Q | K | V along the last axis), sequence b = rows cu_seqlens[b] ..
    cu_seqlens[b + 1] - 1 (int32 [B + 1] on the device, every sequence <= 32 tokens, T = cu_seqlens[-1]: the caller's promise --
    the plan is built on the host, BertModel.pack_plan).  -> [T, H]."""
    qkv = _bf16c(qkv, 'qkv')
    T, H3 = qkv.shape
    if H3 % 3 or (H3 // 3) % heads or (H3 // 3) // heads != 64:
        raise _lib.CreamflHipError(f'bert_attention_varlen: unsupported shape {tuple(qkv.shape)} with {heads} heads')
    if cu_seqlens.dtype != torch.int32 or cu_seqlens.device != qkv.device or not cu_seqlens.is_contiguous():
        raise _lib.CreamflHipError('bert_attention_varlen: cu_seqlens must be a contiguous int32 tensor on the device of qkv')
    return _AttnSmallVarlenFn.apply(qkv, cu_seqlens, int(heads))


def bert_attention_supported(L, head_dim):
    return L <= 32 and head_dim == 64


def bert_attention(qkv, key_mask, heads):
    """softmax(Q K^T / sqrt(d) + key-padding mask) V for a fused projection qkv [B, L, 3H] (Q | K | V along the last
    axis), L <= 32, head dim 64 (csrc/attn_small.hip).  key_mask: [B, L] bool (True = attend) or None.  -> [B, L, H]."""
    qkv = _bf16c(qkv, 'qkv')
    B, L, H3 = qkv.shape
    if H3 % 3 or (H3 // 3) % heads or not bert_attention_supported(L, H3 // 3 // heads):
        raise _lib.CreamflHipError(f'bert_attention: unsupported shape {tuple(qkv.shape)} with {heads} heads')
    m = None
    if key_mask is not None:
        m = key_mask.reshape(B, L)
        if m.dtype != torch.uint8 or m.device != qkv.device or not m.is_contiguous():
            m = m.to(device=qkv.device, dtype=torch.uint8).contiguous()
    return _AttnSmallFn.apply(qkv, m, int(heads))


@torch.no_grad()
def dropout_keep_mask(seed, p, shape, device):
    """The keep mask bert_dropout_add_layernorm uses for (seed, p) on a tensor of this shape (test helper)."""
    lib = _lib.load()
    n = 1
    for d in shape:
        n *= int(d)
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(lib.cfl_dropout_mask(int(seed), float(p), n, _ptr(keep), _stream(keep)), 'cfl_dropout_mask')
    return keep.view(*shape).bool()


# --------------------------------------------------------------------------- A6: retrieval ranks
@torch.no_grad()
def rank_count(q_features, g_features, q_labels, g_labels):
    """A6 (eval_coco.py:37-51,296-317): int32 rank of the best positive for every query."""
    lib = _lib.load()
    Q = _f32(q_features, 'q_features')
    G = _f32(g_features, 'g_features')
    ql = torch.as_tensor(q_labels).to(device=Q.device, dtype=torch.int64).contiguous()
    gl = torch.as_tensor(g_labels).to(device=Q.device, dtype=torch.int64).contiguous()
    if len(Q) != len(ql):
        raise RuntimeError('length mismatch {}, {}'.format(tuple(Q.shape), tuple(ql.shape)))
    if len(G) != len(gl):
        raise RuntimeError('length mismatch {}, {}'.format(tuple(G.shape), tuple(gl.shape)))
    Nq, D = Q.shape
    Ng = G.shape[0]
    ranks = torch.empty(Nq, dtype=torch.int32, device=Q.device)
    ws = _ws(lib.cfl_rank_ws_bytes(Nq, Ng, D), Q.device)
    _lib.check(lib.cfl_rank_count(_ptr(Q), _ptr(G), _ptr(ql), _ptr(gl), Nq, Ng, D, _ptr(ranks), _ptr(ws), _stream(Q)),
               'cfl_rank_count')
    return ranks
