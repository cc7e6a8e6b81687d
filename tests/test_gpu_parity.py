"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden vectors.  Run on an MI355X with `pytest -m gpu`.

Tolerances (north star: loss within 1e-4 fp32, retrieval indices exact):
  * vs. the fp64 closed forms of the oracle: rtol 2e-5 on losses, 1e-4 (of the tensor's scale) on grads
  * vs. the reference's own fp32 numbers (golden): rtol 1e-4 on losses plus the reference's
    documented fp32 cancellation budget (tests/test_oracle_golden.py), 1e-3 of scale on grads
  * ranks / R@K: exact
"""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib
    _lib.load()                       # fail loudly if the extension is missing
    return torch.device('cuda:0')


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _unit(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


def _close(got, want, rtol, atol, msg=''):
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64),
                               rtol=rtol, atol=atol, err_msg=msg)


# ------------------------------------------------------------------------------------------ A1
def _run_pair(dev, I, T, a, b):
    from creamfl_amd import ops
    Ig = I.to(dev).requires_grad_(True)
    Tg = T.to(dev).requires_grad_(True)
    ag = torch.tensor([a], device=dev, requires_grad=True)
    bg = torch.tensor([b], device=dev, requires_grad=True)
    loss, stats = ops.pair_loss(Ig, Tg, ag, bg)
    loss.backward()
    return (loss.item(), stats.cpu().numpy(), Ig.grad.cpu().numpy(), Tg.grad.cpu().numpy(),
            ag.grad.item(), bg.grad.item())


@pytest.mark.parametrize('fname', golden_files('a1_'))
def test_a1_pair_loss_golden(dev, fname):
    z = _load(fname)
    I, T = torch.from_numpy(z['I']), torch.from_numpy(z['T'])
    a, b = float(z['a']), float(z['b'])
    n = I.shape[0]
    loss, stats, dI, dT, da, db = _run_pair(dev, I, T, a, b)
    cf = oracle.pair_loss_closed_form(I, T, a, b)
    g = oracle.pair_loss_grads_closed_form(I, T, a, b)
    # exact (fp64) oracle
    _close(loss, cf['loss'].item(), 2e-5, 1e-6)
    _close(stats[1], cf['pos'].item(), 2e-5, 1e-6)
    _close(stats[2], cf['neg'].item(), 2e-5, 1e-6)
    scale = max(np.abs(g['dI'].numpy()).max(), 1e-12)
    _close(dI, g['dI'].numpy(), 1e-4, 1e-4 * scale, 'dI')
    _close(dT, g['dT'].numpy(), 1e-4, 1e-4 * scale, 'dT')
    _close(da, g['da'].item(), 1e-4, 1e-5)
    _close(db, g['db'].item(), 1e-4, 1e-5)
    # the reference's own fp32 result
    noise = n * n * 2.5e-7
    _close(loss, float(z['loss']), 1e-4, 2 * noise)
    gat = max(1e-3 * np.abs(z['dI']).max(), n * a * 1.2e-7)
    _close(dI, z['dI'], 1e-3, gat, 'dI vs reference')
    _close(dT, z['dT'], 1e-3, gat, 'dT vs reference')


@pytest.mark.parametrize('n,d,matched', [(1, 8, False), (7, 5, False), (64, 64, True), (65, 130, False),
                                         (129, 257, True), (256, 512, True), (300, 768, False),
                                         (2048, 512, True), (2049, 96, False), (2080, 96, True), (2112, 32, False),
                                         # round 6: N % 256 == 0 and D % 256 == 0 take the 256 x 256-tile split-K backward (2048 x 512: 8 splits of 8
                                         # stages; 2304 x 256: 9 row tiles, 8 splits of 9 stages); 2304 x 96 stays on the 128 x 128 ring
                                         (2304, 256, False), (2304, 96, True)])
def test_a1_pair_loss_shapes(dev, n, d, matched):
    """ragged / odd sizes, both tile variants (64x64 below 2048 rows, 128x128 from 2048), unaligned D; from N = 2048 with
    N % 32 == 0 and D % 32 == 0 the IMAGE mode (pre-split operand images, coefficients written in split form, ring-buffered
    backward): full tiles (2048, 512), a partial column tile and an odd tile-row count (2080, 96), a single K stage (2112, 32)."""
    gen = torch.Generator().manual_seed(n * 1000 + d)
    I = _unit(gen, n, d)
    T = torch.nn.functional.normalize(I + 0.5 * _unit(gen, n, d), dim=-1) if matched else _unit(gen, n, d)
    a, b = 15.0, 15.0
    loss, stats, dI, dT, da, db = _run_pair(dev, I, T, a, b)
    if n <= 512:
        cf = oracle.pair_loss_closed_form(I, T, a, b)
        g = oracle.pair_loss_grads_closed_form(I, T, a, b)
    else:   # the O(N^2 D) oracle is too slow: chunked fp64 closed form on the GPU-free path
        cf, g = _closed_form_big(I, T, a, b)
    _close(loss, float(cf['loss']), 3e-5, 1e-6)
    scale = max(float(np.abs(np.asarray(g['dI'])).max()), 1e-12)
    _close(dI, np.asarray(g['dI']), 1e-4, 2e-4 * scale, 'dI')
    _close(dT, np.asarray(g['dT']), 1e-4, 2e-4 * scale, 'dT')
    _close(da, float(g['da']), 2e-4, 1e-5)
    _close(db, float(g['db']), 2e-4, 1e-5)


def _closed_form_big(I, T, a, b, eps=1e-6):
    """fp64 closed form via the GEMM identity on CPU (used only where the O(N^2 D) broadcast oracle
    would take minutes); diagonal handled exactly."""
    I64, T64 = I.double(), T.double()
    n = I.shape[0]
    d2 = (I64 * I64).sum(1)[:, None] + (T64 * T64).sum(1)[None, :] - 2.0 * I64 @ T64.T
    d2[torch.arange(n), torch.arange(n)] = ((I64 - T64) ** 2).sum(1)
    d = torch.sqrt(d2.clamp_min(0) + eps)
    s = -a * d + b
    m = -torch.ones(n, n, dtype=torch.float64)
    m.fill_diagonal_(1.0)
    nll = torch.nn.functional.softplus(-2.0 * m * s)
    gg = 4.0 * m * torch.sigmoid(-2.0 * m * s)
    c = a * gg / d
    return ({'loss': 2.0 * nll.sum()},
            {'dI': (I64 * c.sum(1, keepdim=True) - c @ T64).numpy(),
             'dT': (T64 * c.sum(0)[:, None] - c.t() @ I64).numpy(),
             'da': (gg * d).sum(), 'db': -gg.sum()})


def test_a1_linearity_in_upstream_gradient(dev):
    """size-independent property at the bench size: grads scale linearly with the upstream gradient and
    the loss is symmetric under swapping the two modalities."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(5)
    I = _unit(gen, 256, 512).to(dev)
    T = torch.nn.functional.normalize(I.cpu() + 0.5 * _unit(gen, 256, 512), dim=-1).to(dev)
    a = torch.tensor([15.0], device=dev)
    b = torch.tensor([15.0], device=dev)
    Ig = I.clone().requires_grad_(True)
    l1, _ = ops.pair_loss(Ig, T, a, b)
    (3.0 * l1).backward()
    g3 = Ig.grad.clone()
    Ig2 = I.clone().requires_grad_(True)
    l2, _ = ops.pair_loss(Ig2, T, a, b)
    l2.backward()
    torch.testing.assert_close(g3, 3.0 * Ig2.grad, rtol=1e-6, atol=1e-9)
    Tg = T.clone().requires_grad_(True)
    l3, _ = ops.pair_loss(Tg, I, a, b)          # swapped roles
    l3.backward()
    torch.testing.assert_close(l3, l2, rtol=1e-5, atol=1e-6)
    Tg2 = T.clone().requires_grad_(True)
    l4, _ = ops.pair_loss(I, Tg2, a, b)
    l4.backward()
    torch.testing.assert_close(Tg.grad, Tg2.grad, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------ A3 / A4
def _run_contrast(dev, f, g_same, g_other, d_idx, f_old, w, scale, use_inter=True, use_intra=True, root=False):
    from creamfl_amd.algorithms.contrast import client_contrast_loss
    fg = f.to(dev).requires_grad_(True)
    loss, li, lm = client_contrast_loss(fg, g_same.to(dev), g_other.to(dev), d_idx, f_old.to(dev),
                                        interintra_weight=w, loss_scale=scale, use_inter=use_inter,
                                        use_intra=use_intra, root=root)
    loss.backward()
    return (loss.item(), None if li is None else li.item(), None if lm is None else lm.item(),
            fg.grad.cpu().numpy())


@pytest.mark.parametrize('root', [False, True])
@pytest.mark.parametrize('fname', golden_files('a34_'))
def test_a34_client_contrast_golden(dev, fname, root):
    """root=True: the trainers' form -- the loss is the root of the backward pass, the finish launch writes the final gradient
    (want_grad = 2) and the backward launches nothing; same goldens, same tolerances (--loss_scale goldens take the two-gradient
    form either way)."""
    from functools import partial
    z = _load(fname)
    args = (torch.from_numpy(z['f']), torch.from_numpy(z['g_same']), torch.from_numpy(z['g_other']),
            [int(v) for v in z['d_idx']], torch.from_numpy(z['f_old']))
    _run_contrast = partial(globals()['_run_contrast'], root=root)
    loss, li, lm, df = _run_contrast(dev, *args, float(z['weight']), bool(z['loss_scale']))
    _close(loss, float(z['loss']), 1e-4, 0)
    _close(li, float(z['loss_inter']), 1e-4, 0)
    _close(lm, float(z['loss_moon']), 1e-4, 0)
    sc = np.abs(z['df']).max()
    _close(df, z['df'], 1e-3, 1e-4 * sc)
    _, _, _, df2 = _run_contrast(dev, *args, 1.0, False, use_intra=False)
    _close(df2, z['df_inter_only'], 1e-3, 1e-4 * np.abs(z['df_inter_only']).max())
    _, _, _, df3 = _run_contrast(dev, *args, 1.0, False, use_inter=False)
    _close(df3, z['df_intra_only'], 1e-3, 1e-4 * np.abs(z['df_intra_only']).max())
    # exact oracle
    cf = oracle.client_contrast_grads_closed_form(*args)
    _close(li, cf['loss_inter'].item(), 2e-5, 0)
    _close(lm, cf['loss_moon'].item(), 2e-5, 0)
    _close(df2, cf['d_inter'].numpy(), 1e-4, 2e-5 * np.abs(cf['d_inter'].numpy()).max())
    _close(df3, cf['d_moon'].numpy(), 1e-4, 2e-5 * np.abs(cf['d_moon'].numpy()).max())


def _run_mm_contrast(dev, z_or_args, w, scale, use_inter=True, use_intra=True, root=False):
    from creamfl_amd.algorithms.contrast import mm_client_contrast_loss
    out_img, out_txt, g_img, g_txt, d_idx, old_img, old_txt = z_or_args
    ig, tg = out_img.to(dev).requires_grad_(True), out_txt.to(dev).requires_grad_(True)
    loss, li, lm = mm_client_contrast_loss(ig, tg, g_img.to(dev), g_txt.to(dev), d_idx, old_img.to(dev), old_txt.to(dev),
                                           interintra_weight=w, loss_scale=scale, use_inter=use_inter, use_intra=use_intra,
                                           root=root)
    loss.backward()
    return (loss.item(), None if li is None else li.item(), None if lm is None else lm.item(), ig.grad.cpu().numpy(),
            tg.grad.cpu().numpy())


@pytest.mark.parametrize('root', [False, True])
@pytest.mark.parametrize('fname', golden_files('a34mm_'))
def test_a34_mm_client_contrast_golden(dev, fname, root):
    """The multi-modal client's contrast block (MMClientTrainer.py:164-206 both terms +- --loss_scale, :246-264 intra only,
    :301-308 inter only) through creamfl_amd.algorithms.contrast.mm_client_contrast_loss vs the reference statement sequence
    (a34mm_*.npz): loss terms 1e-4, both feature gradients 1e-3 of scale; duplicate indices, B not a multiple of 16."""
    from functools import partial
    z = _load(fname)
    args = (torch.from_numpy(z['out_img']), torch.from_numpy(z['out_txt']), torch.from_numpy(z['g_img']),
            torch.from_numpy(z['g_txt']), [int(v) for v in z['d_idx']], torch.from_numpy(z['old_img']),
            torch.from_numpy(z['old_txt']))
    _run_mm_contrast = partial(globals()['_run_mm_contrast'], root=root)
    loss, li, lm, di, dt = _run_mm_contrast(dev, args, float(z['weight']), bool(z['loss_scale']))
    _close(loss, float(z['loss']), 1e-4, 0)
    _close(li, float(z['loss_inter']), 1e-4, 0)
    _close(lm, float(z['loss_intra']), 1e-4, 0)
    _close(di, z['d_img'], 1e-3, 1e-4 * np.abs(z['d_img']).max())
    _close(dt, z['d_txt'], 1e-3, 1e-4 * np.abs(z['d_txt']).max())
    loss, li, lm, di, dt = _run_mm_contrast(dev, args, 0.5, False, use_inter=False)
    assert li is None
    _close(loss, float(z['loss_intra_only']), 1e-4, 0)
    _close(di, z['d_img_intra_only'], 1e-3, 1e-4 * np.abs(z['d_img_intra_only']).max())
    _close(dt, z['d_txt_intra_only'], 1e-3, 1e-4 * np.abs(z['d_txt_intra_only']).max())
    loss, li, lm, di, dt = _run_mm_contrast(dev, args, 0.5, False, use_intra=False)
    assert lm is None
    _close(loss, float(z['loss_inter_only']), 1e-4, 0)
    _close(di, z['d_img_inter_only'], 1e-3, 1e-4 * np.abs(z['d_img_inter_only']).max())
    _close(dt, z['d_txt_inter_only'], 1e-3, 1e-4 * np.abs(z['d_txt_inter_only']).max())


@pytest.mark.parametrize('root', [False, True])
@pytest.mark.parametrize('b,m,d,scale', [(128, 50000, 256, False), (128, 50000, 256, True), (50, 7001, 768, True),
                                         (50, 7001, 512, False), (33, 999, 100, False), (5, 40, 18, True)])
def test_a34_mm_client_contrast_shapes(dev, b, m, d, scale, root):
    """Same block at the public-set size (M = 50 000), at d = 768 (column-split wave pairs), at a width the fused kernels do not
    take (d % 4 != 0: the four-op fallback chain), vs the fp32 oracle restatement and the fp64 closed forms of the two
    modalities; the fused path must be 4 forward launches + 1 backward launch -- and, with the loss declared the root of the
    backward pass and no --loss_scale, 4 + 0."""
    from functools import partial
    _run_mm_contrast = partial(globals()['_run_mm_contrast'], root=root)
    from creamfl_amd import _lib, ops
    gen = torch.Generator().manual_seed(b + m + d)
    g_img = _unit(gen, m, d)
    g_txt = torch.nn.functional.normalize(g_img + 0.7 * _unit(gen, m, d), dim=-1)
    d_idx = torch.randint(0, m, (b,), generator=gen).tolist()
    out_img = torch.nn.functional.normalize(g_img[d_idx] + 0.8 * _unit(gen, b, d), dim=-1)
    out_txt = torch.nn.functional.normalize(g_txt[d_idx] + 0.8 * _unit(gen, b, d), dim=-1)
    old_img = torch.nn.functional.normalize(out_img + 0.4 * _unit(gen, b, d), dim=-1)
    old_txt = torch.nn.functional.normalize(out_txt + 0.4 * _unit(gen, b, d), dim=-1)
    # the banks live on the device for the whole round (MMClientTrainer.train_epoch): their images are built once, not per step
    args = (out_img, out_txt, g_img.to(dev), g_txt.to(dev), d_idx, old_img, old_txt)
    w = 0.5
    fused = ops.bank_attn_supported(b, m, d)
    assert fused == (d % 4 == 0)
    _run_mm_contrast(dev, args, w, scale)                         # builds the bank images (not a step launch)
    _lib.prof_enable(True)
    _lib.prof_reset()
    loss, li, lm, di, dt = _run_mm_contrast(dev, args, w, scale)
    torch.cuda.synchronize()
    launches = {k: v[0] for k, v in _lib.prof_query().items()}
    _lib.prof_enable(False)
    if fused:
        assert sum(launches.values()) == (4 if (root and not scale) else 5), launches
    ci = oracle.client_contrast_grads_closed_form(out_img, g_img, g_txt, d_idx, old_img)
    ct = oracle.client_contrast_grads_closed_form(out_txt, g_txt, g_img, d_idx, old_txt)
    inter = (ci['loss_inter'] + ct['loss_inter']).item()
    intra = ((ci['loss_moon'] + ct['loss_moon']) / 2).item()
    ratio = inter / intra if scale else 1.0
    _close(li, inter, 2e-5, 1e-6)
    _close(lm, intra, 2e-5, 1e-6)
    _close(loss, (intra + inter / ratio) * w, 1e-4, 0)
    for got, c in ((di, ci), (dt, ct)):
        want = (c['d_moon'].numpy() / 2 + c['d_inter'].numpy() / ratio) * w
        _close(got, want, 2e-4, 5e-5 * np.abs(want).max())
    if m <= 8000:
        ig, tg = out_img.clone().requires_grad_(True), out_txt.clone().requires_grad_(True)
        ol, oli, olm = oracle.mm_client_contrast_loss(ig, tg, g_img, g_txt, d_idx, old_img, old_txt, interintra_weight=w,
                                                      loss_scale=scale)
        ol.backward()
        _close(loss, ol.item(), 1e-4, 0)
        _close(di, ig.grad.numpy(), 1e-3, 1e-4 * np.abs(ig.grad.numpy()).max())
        _close(dt, tg.grad.numpy(), 1e-3, 1e-4 * np.abs(tg.grad.numpy()).max())


def test_a34_image_cache_follows_the_bank(dev):
    """The pre-split bank image (csrc/bank_gsplit.h) is built once per bank VERSION: repeated steps against the same bank reuse
    it, an in-place update of the bank (a new round's global features written into the same storage) rebuilds it, and the
    result after the update is the oracle's for the NEW bank -- never a stale image."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(77)
    m, d, b = 2000, 128, 40
    G = _unit(gen, m, d).to(dev)
    Gs = _unit(gen, m, d).to(dev)
    idx = torch.randint(0, m, (b,), generator=gen).tolist()
    f = _unit(gen, b, d)
    fo = _unit(gen, b, d)

    def run():
        fg = f.to(dev).requires_grad_(True)
        loss, li, lm, _, _ = ops.client_contrast_fused(fg, Gs, G, idx, fo.to(dev), 0.5, weight=0.5)
        loss.backward()
        return loss.item(), fg.grad.cpu().numpy()

    n0 = ops.BANK_IMAGE_BUILDS[0]
    l1, g1 = run()
    l2, g2 = run()
    assert ops.BANK_IMAGE_BUILDS[0] == n0 + 1, 'second step on the same bank must reuse the image'
    assert l1 == l2 and np.array_equal(g1, g2)
    cf = oracle.client_contrast_grads_closed_form(f, Gs.cpu(), G.cpu(), idx, fo)
    want = (cf['loss_moon'].item() + cf['loss_inter'].item()) * 0.5
    _close(l1, want, 2e-5, 0)
    with torch.no_grad():
        G.copy_(_unit(gen, m, d).to(dev))                     # in-place: same storage, new version
    l3, g3 = run()
    assert ops.BANK_IMAGE_BUILDS[0] == n0 + 2, 'an in-place update of the bank must rebuild the image'
    cf = oracle.client_contrast_grads_closed_form(f, Gs.cpu(), G.cpu(), idx, fo)
    want3 = (cf['loss_moon'].item() + cf['loss_inter'].item()) * 0.5
    _close(l3, want3, 2e-5, 0)
    assert abs(l3 - l1) > 1e-3 * abs(l1)
    wg = (cf['d_moon'].numpy() + cf['d_inter'].numpy()) * 0.5
    _close(g3, wg, 1e-4, 3e-5 * np.abs(wg).max())


@pytest.mark.parametrize('b,m,d', [(128, 50000, 256), (96, 7001, 512), (40, 3000, 768), (33, 1500, 64)])
def test_a34_image_path_equals_fp32_bank_path(dev, b, m, d, monkeypatch):
    """Same step through the single-pass bank kernels (pre-split image, 16-row slots; column-split wave pairs beyond D = 256)
    and through the ONE reference path kept beside them, the exact-fp32 two-pass kernels of csrc/bank.hip (CFL_BANK_EXACT):
    same loss terms and gradients within the 3 x bf16-split error (both are separately compared with the oracle elsewhere)."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(b + m + d)
    G, Gs = _unit(gen, m, d), _unit(gen, m, d)
    idx = torch.randint(0, m, (b,), generator=gen).tolist()
    f = torch.nn.functional.normalize(Gs[idx] + 0.9 * _unit(gen, b, d), dim=-1)
    fo = torch.nn.functional.normalize(f + 0.4 * _unit(gen, b, d), dim=-1)
    got = _run_contrast(dev, f, Gs, G, idx, fo, 0.5, True)
    monkeypatch.setattr(ops, '_BANK_EXACT', True)
    assert not ops.bank_attn_supported(b, m, d)
    ref = _run_contrast(dev, f, Gs, G, idx, fo, 0.5, True)
    for a, r in zip(got[:3], ref[:3]):
        _close(a, r, 2e-5, 0)
    _close(got[3], ref[3], 1e-4, 3e-5 * np.abs(ref[3]).max())


@pytest.mark.parametrize('root', [False, True])
@pytest.mark.parametrize('m,d', [(9, 32), (4097, 256), (300, 768)])
def test_a34_duplicate_and_boundary_indices(dev, m, d, root):
    """collisions in the batch's public-set indices (the same representation is the positive of several rows), the first and the
    last bank row as positives, inter + intra terms with and without loss_scale: == the closed-form oracle."""
    from functools import partial
    _run_contrast = partial(globals()['_run_contrast'], root=root)
    gen = torch.Generator().manual_seed(m + d)
    g_same, g_other = _unit(gen, m, d), _unit(gen, m, d)
    d_idx = [0, m - 1, 0, 0, m - 1, m // 2, m // 2, 1, m - 2, 0, m - 1, 3 % m]
    f = torch.nn.functional.normalize(g_same[d_idx] + 0.8 * _unit(gen, len(d_idx), d), dim=-1)
    f_old = torch.nn.functional.normalize(f + 0.3 * _unit(gen, len(d_idx), d), dim=-1)
    args = (f, g_same, g_other, d_idx, f_old)
    cf = oracle.client_contrast_grads_closed_form(*args)
    _, li, lm, _ = _run_contrast(dev, *args, 0.5, False)
    _close(li, cf['loss_inter'].item(), 2e-5, 1e-6)
    _close(lm, cf['loss_moon'].item(), 2e-5, 1e-6)
    _, _, _, di = _run_contrast(dev, *args, 1.0, False, use_intra=False)
    _, _, _, dm = _run_contrast(dev, *args, 1.0, False, use_inter=False)
    _close(di, cf['d_inter'].numpy(), 1e-4, 3e-5 * np.abs(cf['d_inter'].numpy()).max())
    _close(dm, cf['d_moon'].numpy(), 1e-4, 3e-5 * np.abs(cf['d_moon'].numpy()).max())
    # both terms, plain combination (the direct form when root): (loss_moon + loss_inter) * w
    w = 0.25
    loss, _, _, df = _run_contrast(dev, *args, w, False)
    _close(loss, (cf['loss_moon'].item() + cf['loss_inter'].item()) * w, 1e-4, 0)
    want = (cf['d_moon'].numpy() + cf['d_inter'].numpy()) * w
    _close(df, want, 2e-4, 5e-5 * np.abs(want).max())
    # both terms with loss_scale: (loss_moon + loss_inter / (loss_inter / loss_moon).detach()) * w  (ClientTrainer.py:416-419)
    loss, li2, lm2, df = _run_contrast(dev, *args, w, True)
    ratio = cf['loss_inter'].item() / cf['loss_moon'].item()
    _close(loss, (cf['loss_moon'].item() + cf['loss_inter'].item() / ratio) * w, 1e-4, 0)
    want = (cf['d_moon'].numpy() + cf['d_inter'].numpy() / ratio) * w
    _close(df, want, 2e-4, 5e-5 * np.abs(want).max())


@pytest.mark.parametrize('b,m,d', [(128, 50000, 256), (256, 20000, 512), (37, 900, 768), (130, 3000, 128)])
def test_a34_root_form_equals_the_two_launch_form(dev, b, m, d):
    """The direct finish (final gradient for an upstream gradient of 1, no backward launch) against the unit gradients + backward
    launch of the same step: same loss bit for bit (the forward is the same code), gradients to fp32 rounding of the last
    combination; 2 launches instead of 3; a scaled loss (root=False) still gets its scale."""
    from creamfl_amd import _lib, ops
    gen = torch.Generator().manual_seed(b + m + d)
    G, Gs = _unit(gen, m, d).to(dev), _unit(gen, m, d).to(dev)
    idx = torch.randint(0, m, (b,), generator=gen).tolist()
    f = torch.nn.functional.normalize(Gs.cpu()[idx] + 0.9 * _unit(gen, b, d), dim=-1)
    fo = torch.nn.functional.normalize(f + 0.4 * _unit(gen, b, d), dim=-1).to(dev)

    def run(root, scale_by=None, **kw):
        fg = f.to(dev).requires_grad_(True)
        loss = ops.client_contrast_fused(fg, Gs, G, idx, fo, 0.5, weight=0.3, root=root, **kw)[0]
        (loss if scale_by is None else loss * scale_by).backward()
        return loss.item(), fg.grad

    for kw in ({}, {'use_intra': False}, {'use_inter': False}):
        run(True, **kw)                                                 # bank image built, modules loaded
        _lib.prof_enable(True)
        _lib.prof_reset()
        l1, g1 = run(True, **kw)
        torch.cuda.synchronize()
        n1 = sum(v[0] for v in _lib.prof_query().values())
        _lib.prof_reset()
        l0, g0 = run(False, **kw)
        torch.cuda.synchronize()
        n0 = sum(v[0] for v in _lib.prof_query().values())
        _lib.prof_enable(False)
        assert l1 == l0
        assert n0 - n1 == 1 and n1 == (1 if kw.get('use_inter') is False else 2), (n0, n1)
        torch.testing.assert_close(g1, g0, rtol=2e-6, atol=1e-7 * float(g0.abs().max()))
        _, g3 = run(False, scale_by=3.0, **kw)
        torch.testing.assert_close(g3, 3.0 * g0, rtol=2e-6, atol=1e-7 * float(g0.abs().max()))


@pytest.mark.parametrize('b,m,d', [(1, 1, 4), (3, 200, 17), (64, 4097, 256), (65, 5000, 100), (128, 50000, 256),
                                   (256, 20000, 512), (130, 3000, 768)])
def test_a3_inter_shapes(dev, b, m, d):
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(b * 7 + m)
    G = _unit(gen, m, d)
    idx = torch.randint(0, m, (b,), generator=gen)
    f = torch.nn.functional.normalize(G[idx] + 0.7 * _unit(gen, b, d), dim=-1)
    fg = f.to(dev).requires_grad_(True)
    loss, lse, pos = ops.inter_contrast(fg, G.to(dev), idx.tolist(), 0.5)
    loss.backward()
    cf = oracle.client_contrast_grads_closed_form(f, G, G, idx.tolist(), f)
    # atol: the log-sum-exp uses the 3 x bf16-split logits (|error| ~ 2e-6 at |logit| <= 2), the positive dot is exact fp32;
    # their difference is the whole loss when it is ~0 (M = 1)
    _close(loss.item(), cf['loss_inter'].item(), 2e-5, 1e-5)
    _close(lse.cpu().numpy(), cf['lse'].numpy(), 1e-5, 1e-5)
    _close(pos.cpu().numpy(), cf['pos_inter'].numpy(), 1e-5, 1e-5)
    dref = cf['d_inter'].numpy()
    # atol floor: softmax . G runs on G = hi + lo (two bf16, residual 2^-17 |G|); it only shows where the gradient itself
    # cancels to ~0 (M = 1: softmax = onehot = the target)
    _close(fg.grad.cpu().numpy(), dref, 1e-4, 3e-5 * np.abs(dref).max() + 2e-5 / (0.5 * b))


def test_a3_online_lse_rescale_branch(dev):
    """force the running-max update: one bank row far above the rest, late in the stream, for some
    feature rows only (cdna guide rule 26: a rare data-dependent branch needs its own test)."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(11)
    b, m, d = 96, 9000, 64
    G = 0.1 * torch.randn(m, d, generator=gen)
    f = torch.randn(b, d, generator=gen)
    G[8000] = 6.0 * f[5] / f[5].norm()       # huge logit for row 5 in a late chunk
    G[10] = 4.0 * f[70] / f[70].norm()       # and an early one for row 70
    idx = torch.randint(0, m, (b,), generator=gen)
    loss, lse, _ = ops.inter_contrast(f.to(dev), G.to(dev), idx.tolist(), 0.5)
    want = torch.logsumexp((f.double() @ G.double().T) / 0.5, dim=1)
    _close(lse.cpu().numpy(), want.numpy(), 1e-5, 1e-5)


# ------------------------------------------------------------------------------------------ A5
@pytest.mark.parametrize('fname', golden_files('a5_'))
def test_a5_conw_golden(dev, fname):
    from creamfl_amd import ops
    z = _load(fname)
    vecs = [torch.from_numpy(v).to(dev) for v in z['vecs']]
    G = torch.from_numpy(z['g_other']).to(dev)
    lp = torch.stack([ops.conw_logprob(v, G) for v in vecs], 0)
    agg, w = ops.conw_combine(vecs, lp, return_weights=True)
    _close(lp.cpu().numpy(), z['logprob'], 1e-5, 1e-5)
    _close(w.cpu().numpy(), z['weights'], 1e-4, 1e-6)
    _close(agg.cpu().numpy(), z['agg'], 1e-4, 1e-6)


@pytest.mark.parametrize('m,d,row0,rows', [(1000, 64, 0, 1000), (1000, 64, 300, 333), (5000, 256, 4096, 904),
                                           (2500, 100, 0, 2500)])
def test_a5_conw_row_shards(dev, m, d, row0, rows):
    """rows are independent: any [row0, row0+rows) shard equals the same slice of the full result."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(m + d)
    G = _unit(gen, m, d)
    V = torch.nn.functional.normalize(G + 0.5 * _unit(gen, m, d), dim=-1)
    got = ops.conw_logprob(V.to(dev), G.to(dev), row0, rows).cpu()
    want = oracle.conw_logprob(V.double(), G.double(), literal=False)[row0:row0 + rows]
    _close(got.numpy(), want.numpy(), 1e-5, 1e-5)


@pytest.mark.parametrize('m,d,row0,rows', [(4096, 256, 0, 4096), (3000, 200, 1024, 1500), (2600, 128, 0, 2600),
                                           (1554, 36, 16, 1000), (5000, 256, 4096, 904),
                                           # round 6: 256 < D <= 512 on the 4-wave form (128 rows per workgroup, one wave per SIMD)
                                           (4096, 512, 0, 4096), (3000, 384, 1024, 1500), (2200, 300, 7, 1111), (5000, 512, 4096, 904),
                                           # D <= 768: 16-row steps on the 16 x 16 x 32 MFMA, per-lane running log-sum-exp
                                           (4096, 768, 0, 4096), (3000, 640, 1024, 1500), (2203, 516, 7, 1111), (5000, 768, 4096, 904)])
def test_a5_conw_bank_pass_equals_tile_gemm(dev, m, d, row0, rows, monkeypatch):
    """Round 4: the con_w log-probabilities on the bank pass of rows A3 / A4 (pre-split image of G, 256 rows of V per workgroup
    in registers, online log-sum-exp, exact fp32 positives) against the fp64 oracle (MMFL.py:304-307) and against the tile GEMM of
    bank.hip; the profiler must have seen the bank pass, not the GEMM, and one image build serves every client of the round."""
    from creamfl_amd import _lib, ops
    gen = torch.Generator().manual_seed(3 * m + d)
    G = _unit(gen, m, d)
    V = torch.nn.functional.normalize(G + 0.5 * _unit(gen, m, d), dim=-1)
    V[row0 + 5] = 3.0 * G[row0 + 700]            # a dominant logit far from the diagonal (late slot), and an early one
    V[row0 + rows - 1] = 2.5 * G[3]
    Gd, Vd = G.to(dev), V.to(dev)
    ops.invalidate_bank_images()
    builds0 = ops.BANK_IMAGE_BUILDS[0]
    _lib.prof_enable(True)
    _lib.prof_reset()
    got = ops.conw_logprob(Vd, Gd, row0, rows)
    got2 = ops.conw_logprob(Vd, Gd, row0, rows)                       # a second client against the same bank
    launches = {k: v[0] for k, v in _lib.prof_query().items()}
    _lib.prof_enable(False)
    assert launches.get('cfl_bank_stream_kernel', 0) == 2 and ops.BANK_IMAGE_BUILDS[0] == builds0 + 1, launches
    assert torch.equal(got, got2)
    want = oracle.conw_logprob(V.double(), G.double(), literal=False)[row0:row0 + rows]
    _close(got.cpu().numpy(), want.numpy(), 1e-5, 1e-5)
    monkeypatch.setattr(ops, '_CONW_NOIMG', True)
    ref = ops.conw_logprob(Vd, Gd, row0, rows)
    _close(got.cpu().numpy(), ref.cpu().numpy(), 1e-5, 1e-5)


def test_a5_conw_small_shards_stay_on_the_tile_gemm(dev):
    """Below 512 rows (a shard of a small public set) the bank pass is refused and conw_logprob runs the tile GEMM."""
    from creamfl_amd import _lib, ops
    lib = _lib.load()
    assert lib.cfl_conw_img_supported(511, 5000, 256) == 0 and lib.cfl_conw_img_supported(512, 5000, 256) == 1
    assert lib.cfl_conw_img_supported(4096, 4096, 772) == 0 and lib.cfl_conw_img_supported(4096, 4096, 254) == 0 and lib.cfl_conw_img_supported(4096, 4096, 768) == 1
    assert lib.cfl_conw_img_supported(4096, 4096, 512) == 1 and lib.cfl_conw_img_supported(4096, 4096, 260) == 1
    gen = torch.Generator().manual_seed(5)
    G = _unit(gen, 1000, 64)
    _lib.prof_enable(True)
    _lib.prof_reset()
    ops.conw_logprob(G.to(dev), G.to(dev), 300, 333)
    launches = {k: v[0] for k, v in _lib.prof_query().items()}
    _lib.prof_enable(False)
    assert 'cfl_bank_stream_kernel' not in launches, launches


# ------------------------------------------------------------------------------------------ A2-head
NAMES = ['attention__w_1__weight', 'attention__w_2__weight', 'fc__weight', 'fc__bias',
         'layer_norm__weight', 'layer_norm__bias']


@pytest.mark.parametrize('fused', [True, False], ids=['single_pass', 'three_pass'])
@pytest.mark.parametrize('fname', golden_files('a2_'))
def test_a2_pie_head_golden(dev, fname, fused, monkeypatch):
    """both implementations of the attention pooling (csrc/pie_fused.hip where the widths allow, csrc/pie.hip) against the
    reference's PIENet outputs and gradients."""
    from creamfl_amd.networks.models.pie_model import PIENet
    from creamfl_amd import ops
    monkeypatch.setattr(ops, 'PIE_FUSED', fused)
    z = _load(fname)
    cd, d, dh = z['x'].shape[2], z['out'].shape[1], z['p_attention__w_1__weight'].shape[0]
    net = PIENet(1, cd, d, dh).to(dev)
    sd = {n.replace('__', '.'): torch.from_numpy(z['p_' + n]) for n in NAMES}
    net.load_state_dict(sd)
    x = torch.from_numpy(z['x']).to(dev).requires_grad_(True)
    out = torch.from_numpy(z['out']).to(dev).requires_grad_(True)
    mask = torch.from_numpy(z['mask']).to(dev) if z['mask'].size else None
    o, attn, res = net(out, x, mask)
    y = ops.l2_normalize(o)
    (y * torch.from_numpy(z['gy']).to(dev)).sum().backward()
    _close(o.detach().cpu().numpy(), z['o'], 1e-4, 1e-5)
    _close(attn.detach().cpu().numpy(), z['attn'], 1e-4, 1e-6)
    _close(res.detach().cpu().numpy(), z['res'], 1e-4, 1e-6)
    _close(y.detach().cpu().numpy(), z['y'], 1e-4, 1e-6)
    _close(x.grad.cpu().numpy(), z['dx'], 1e-3, 1e-4 * np.abs(z['dx']).max(), 'dx')
    _close(out.grad.cpu().numpy(), z['dout'], 1e-3, 1e-4 * np.abs(z['dout']).max(), 'dout')
    for n, p in zip(NAMES, [net.attention.w_1.weight, net.attention.w_2.weight, net.fc.weight, net.fc.bias,
                            net.layer_norm.weight, net.layer_norm.bias]):
        ref = z['g_' + n]
        _close(p.grad.cpu().numpy(), ref, 2e-3, 2e-4 * max(np.abs(ref).max(), 1e-8), n)


@pytest.mark.parametrize('n,p,cd,dh,d', [(2, 49, 2048, 1024, 256), (5, 49, 512, 256, 512), (3, 7, 300, 150, 256),
                                         (4, 33, 130, 66, 100)])
def test_a2_fused_image_head_vs_oracle(dev, n, p, cd, dh, d):
    """the fused path used by EncoderImage (avgpool + pool + epilogue incl. l2norm) vs oracle.image_head_glue
    semantics on a [N, P, Cd] map, forward and backward."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(cd + d)
    X = torch.randn(n, p, cd, generator=gen)
    w1 = torch.randn(dh, cd, generator=gen) / cd ** 0.5
    w2 = torch.randn(1, dh, generator=gen) / dh ** 0.5
    fcw = torch.randn(d, cd, generator=gen) / cd ** 0.5
    fcb = 0.1 * torch.randn(d, generator=gen)
    pfw = torch.randn(d, cd, generator=gen) / cd ** 0.5
    pfb = 0.1 * torch.randn(d, generator=gen)
    lnw = 1 + 0.1 * torch.randn(d, generator=gen)
    lnb = 0.1 * torch.randn(d, generator=gen)
    gy = torch.randn(n, d, generator=gen)

    def run(dv, fused):
        t = [v.to(dv).requires_grad_(True) for v in (X, w1, w2, fcw, fcb, pfw, pfb, lnw, lnb)]
        Xd, w1d, w2d, fcwd, fcbd, pfwd, pfbd, lnwd, lnbd = t
        if fused:
            H = torch.nn.functional.linear(Xd, w1d)
            pooled, attn, xmean = ops.pie_pool(Xd, H, w2d, None, want_mean=True)
            out = torch.nn.functional.linear(xmean, fcwd, fcbd)
            y, o, r = ops.pie_epilogue(out, torch.nn.functional.linear(pooled, pfwd, pfbd), lnwd, lnbd)
        else:
            out = torch.nn.functional.linear(Xd.mean(1), fcwd, fcbd)
            o, attn, r = oracle.pie_head(out, Xd, w1d, w2d, pfwd, pfbd, lnwd, lnbd)
            y = oracle.l2_normalize(o)
        (y * gy.to(dv)).sum().backward()
        return [y.detach().cpu().numpy()] + [v.grad.cpu().numpy() for v in t]

    got = run(dev, True)
    want = run(torch.device('cpu'), False)
    for i, (g_, w_) in enumerate(zip(got, want)):
        _close(g_, w_, 2e-3, 2e-4 * max(np.abs(w_).max(), 1e-8), f'tensor {i}')


@pytest.mark.parametrize('n,p,cd,dh,masked', [(256, 49, 2048, 1024, False), (5, 49, 512, 256, False), (7, 12, 304, 152, True),
                                              (3, 1000, 64, 64, True), (40, 49, 768, 384, False), (2, 5, 8, 8, False)])
def test_a2_pool_bf16_operands_single_pass(dev, n, p, cd, dh, masked, monkeypatch):
    """autocast regime: bf16 X / H go through the single-pass kernels unconverted; result == the fp32 three-pass kernels
    on the same (bf16-valued) operands, dX / dH rounded to bf16 once."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(n + cd)
    X = torch.randn(n, p, cd, generator=gen).bfloat16()
    H = (torch.randn(n, p, dh, generator=gen) * 1.5).bfloat16()
    w2 = torch.randn(dh, generator=gen) / dh ** 0.5
    gp = torch.randn(n, cd, generator=gen)
    gm = torch.randn(n, cd, generator=gen)
    mask = None
    if masked:
        lens = torch.randint(1, p + 1, (n,), generator=gen)
        mask = (torch.arange(p)[None, :] >= lens[:, None]).to(dev)

    def run(fused, dtype):
        monkeypatch.setattr(ops, 'PIE_FUSED', fused)
        x = X.to(dev, dtype).requires_grad_(True)
        h = H.to(dev, dtype).requires_grad_(True)
        w = w2.to(dev).requires_grad_(True)
        pooled, attn, xmean = ops.pie_pool(x, h, w, mask, want_mean=True)
        ((pooled * gp.to(dev)).sum() + (xmean * gm.to(dev)).sum()).backward()
        return [t.detach().float().cpu().numpy() for t in (pooled, attn, xmean, x.grad, h.grad, w.grad)], x.grad.dtype

    assert ops._lib.load().cfl_pie_fused_supported(n, p, cd, dh, 1) == 1
    got, gdt = run(True, torch.bfloat16)
    assert gdt == torch.bfloat16
    want, _ = run(False, torch.float32)
    f32, _ = run(True, torch.float32)
    names = ['pooled', 'attn', 'xmean', 'dX', 'dH', 'dw2']
    for nm, g_, w_, f_ in zip(names, got, want, f32):
        scale = max(float(np.abs(w_).max()), 1e-8)
        _close(f_, w_, 1e-4, 2e-6 * scale, nm + ' (fp32 single pass vs three pass)')
        if nm in ('dX', 'dH'):                               # one bf16 rounding (2^-8 relative)
            _close(g_, w_, 4.5e-3, 1e-5 * scale, nm + ' (bf16)')
        else:
            _close(g_, w_, 1e-4, 2e-6 * scale, nm + ' (bf16 operands)')


# ------------------------------------------------------------------------------------------ A6
@pytest.mark.parametrize('fname', golden_files('a6_'))
def test_a6_recall_golden(dev, fname):
    from creamfl_amd import ops
    z = _load(fname)
    keys = [str(k) for k in z['keys']]
    for (q, g, ql, gl, want) in [(z['img'], z['cap'], z['img_cls'], z['cap_cls'], z['i2t']),
                                 (z['cap'], z['img'], z['cap_cls'], z['img_cls'], z['t2i'])]:
        ranks = ops.rank_count(torch.from_numpy(q).to(dev), torch.from_numpy(g).to(dev), ql, gl).cpu().numpy()
        ref = oracle.recall_ranks_count(q, g, ql, gl)
        assert np.array_equal(ranks.astype(np.float64), ref)           # indices exact
        sc = oracle.recall_scores(ranks.astype(np.float64))
        np.testing.assert_array_equal(np.array([sc[k] for k in keys]), want)   # == reference evaluator


@pytest.mark.parametrize('nq,ng,d', [(130, 257, 50), (1, 1, 4), (300, 1000, 64), (129, 1, 16), (64, 4100, 33)])
def test_a6_ragged_no_positive_duplicates(dev, nq, ng, d):
    """tile edges (sizes off the 128 x 128 x 16 grid, D not a multiple of 4: scalar fetch), queries without any positive
    (rank = gallery size), several positives per query scattered over tiles, exact duplicates of the best positive
    (strict >: not counted), label values shared by far-apart gallery rows: ranks == the fp64 count oracle, exactly."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(nq * 7 + ng)
    q = _unit(gen, nq, d)
    g = _unit(gen, ng, d)
    ql = torch.arange(nq) % 97
    gl = torch.randint(0, 97, (ng,), generator=gen)
    gl[gl == 5] = 1000                                  # queries labelled 5: no positive anywhere
    if ng >= 8:
        g[ng - 1] = g[0]                                # duplicated rows (also duplicates of some query's best positive)
        g[ng // 2] = g[1]
        gl[ng - 1] = gl[0]
    ranks = ops.rank_count(q.to(dev), g.to(dev), ql, gl).cpu().numpy()
    want = oracle.recall_ranks_count(q.numpy(), g.numpy(), ql.numpy(), gl.numpy())
    assert np.array_equal(ranks.astype(np.float64), want)
    nopos = ~np.isin(ql.numpy(), gl.numpy())
    assert (ranks[nopos] == ng).all()


def test_a6_coco_1k_fold_size(dev):
    """one 1K fold (1000 images x 5000 captions, D = 512): exact ranks vs the fp64 count oracle."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(3)
    img = _unit(gen, 1000, 512)
    cap = torch.nn.functional.normalize(img.repeat_interleave(5, 0) + 1.8 * _unit(gen, 5000, 512), dim=-1)
    icls, ccls = np.arange(1000), np.arange(5000) // 5
    for (q, g, ql, gl) in [(img, cap, icls, ccls), (cap, img, ccls, icls)]:
        ranks = ops.rank_count(q.to(dev), g.to(dev), ql, gl).cpu().numpy()
        assert np.array_equal(ranks.astype(np.float64), oracle.recall_ranks_count(q.numpy(), g.numpy(), ql, gl))


def test_a6_coco_5k_protocol_size(dev):
    """The 5K protocol of the evaluator (eval_coco.py:392-448: 5000 images x 25 000 captions, no folds) at the server
    dimension d = 512, both directions: every rank equal to the fp64 count oracle (chunked on the host).  Synthetic
    features with a noise level that spreads the ranks over a wide range (R@1 well below 100)."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(55)
    img = _unit(gen, 5000, 512)
    cap = torch.nn.functional.normalize(img.repeat_interleave(5, 0) + 8.0 * _unit(gen, 25000, 512), dim=-1)
    icls, ccls = np.arange(5000), np.arange(25000) // 5
    for (q, g, ql, gl) in [(img, cap, icls, ccls), (cap, img, ccls, icls)]:
        ranks = ops.rank_count(q.to(dev), g.to(dev), ql, gl).cpu().numpy().astype(np.float64)
        want = oracle.recall_ranks_count(q.numpy(), g.numpy(), ql, gl, block=500)
        assert np.array_equal(ranks, want), int((ranks != want).sum())
        assert 1.0 < 100.0 * float((want < 1).mean()) < 99.0         # a non-trivial ranking


# ------------------------------------------------------------------------------------------ KD term (8f-1)
@pytest.mark.parametrize('b,m,d,w', [(1, 3, 4, 1.0), (37, 500, 100, 0.3), (128, 50000, 256, 0.3), (256, 1000, 512, 2.0)])
def test_kd_mse_matches_torch(dev, b, m, d, w):
    """kd_weight * nn.MSELoss()(out, agg[d_idx]) (src/algorithms/MMFL.py:352-378): value and gradient."""
    from creamfl_amd import ops
    gen = torch.Generator().manual_seed(b + m)
    agg = torch.randn(m, d, generator=gen)
    out = torch.randn(b, d, generator=gen)
    idx = torch.randint(0, m, (b,), generator=gen)
    og = out.to(dev).requires_grad_(True)
    loss = ops.kd_mse(og, agg.to(dev), idx.tolist(), w)
    (2.5 * loss).backward()
    oc = out.double().requires_grad_(True)
    ref = w * torch.nn.MSELoss()(oc, agg.double()[idx])
    (2.5 * ref).backward()
    _close(loss.item(), ref.item(), 1e-5, 0)
    _close(og.grad.cpu().numpy(), oc.grad.numpy(), 1e-5, 1e-9)


# ------------------------------------------------------------------------------------------ supervised glue (8f-4)
@pytest.mark.parametrize('fname', golden_files('f4_'))
def test_f4_supervised_glue_golden(dev, fname):
    """cfl_sup_glue_fwd/bwd against the reference-generated vectors (ClientTrainer.py:344-357)."""
    from creamfl_amd import ops
    z = _load(fname)
    fvec = torch.from_numpy(z['fvec']).to(dev).requires_grad_(True)
    W = torch.from_numpy(z['class_weight']).to(dev).requires_grad_(True)
    labels = torch.from_numpy(z['labels']).to(dev)
    total, stats = ops.supervised_glue(fvec, labels, W, float(z['margin']), int(z['topk']))
    total.backward()
    st = stats.cpu().numpy()
    _close(total.item(), float(z['total']), 1e-5, 0)
    _close(st[1], float(z['ce']), 1e-5, 0)
    _close(st[2], float(z['center']), 1e-5, 0)
    assert st[3] == np.float32(z['prec1'][0]) and st[4] == np.float32(z['preck'][0])      # counts: exact
    _close(fvec.grad.cpu().numpy(), z['dfvec'], 1e-4, 1e-8)
    _close(W.grad.cpu().numpy(), z['dclass_weight'], 1e-4, 1e-8)


@pytest.mark.parametrize('b,c,dw,k', [(1, 2, 1, 1), (512, 100, 512, 5), (513, 80, 2048, 5), (64, 1000, 300, 7), (3, 4096, 8, 2)])
def test_f4_supervised_glue_vs_oracle(dev, b, c, dw, k):
    """Shapes beyond the goldens (batch 512 of the reference, ragged C / Dw, the C limit), fp64 oracle."""
    from creamfl_amd import ops
    from oracle import supervised
    gen = torch.Generator().manual_seed(b * 7 + c)
    labels = torch.randint(0, c, (b,), generator=gen)
    fv = torch.randn(b, c, generator=gen) * 3
    W = torch.relu(torch.randn(c, dw, generator=gen) * (2.0 / dw ** 0.5))
    fg = fv.to(dev).requires_grad_(True)
    Wg = W.to(dev).requires_grad_(True)
    total, stats = ops.supervised_glue(fg, labels.to(dev), Wg, 4.0, k)
    (1.7 * total).backward()
    fc = fv.double().requires_grad_(True)
    Wc = W.double().requires_grad_(True)
    t, ce, cen, p1, pk = supervised.supervised_glue(fc, labels, Wc, 4.0, k)
    (1.7 * t).backward()
    st = stats.cpu().numpy()
    _close(total.item(), t.item(), 2e-5, 0)
    _close(st[:3], [t.item(), ce.item(), cen.item()], 2e-5, 0)
    _close(st[3:], [float(p1), float(pk)], 1e-6, 0)
    _close(fg.grad.cpu().numpy(), fc.grad.numpy(), 1e-4, 1e-8)
    _close(Wg.grad.cpu().numpy(), Wc.grad.numpy(), 1e-4, 1e-8 + 1e-5 * float(Wc.grad.abs().max()))


def test_f4_supervised_glue_bad_label_is_loud(dev):
    from creamfl_amd import ops
    fv = torch.randn(4, 3, device=dev)
    W = torch.rand(3, 8, device=dev)
    total, _ = ops.supervised_glue(fv, torch.tensor([0, 1, 3, 2], device=dev), W, 4.0, 2)
    assert torch.isnan(total)


# ------------------------------------------------------------------------------------------ bf16 GEMM probe
@pytest.mark.parametrize('m,n,k,variant', [(256, 64, 64, 0), (1000, 128, 256, 0), (50176, 1024, 256, 22), (12544, 512, 2048, 0),
                                           (777, 72, 128, 21), (4096, 256, 1024, 44), (3000, 256, 512, 42), (513, 64, 320, 41)])
def test_gemm_bf16_nt_matches_torch(dev, m, n, k, variant):
    """cfl_gemm_bf16_nt vs an fp32 matmul of the same bf16 inputs (asymmetric random operands, ragged M / N);
    tolerance = one bf16 rounding of the output."""
    from creamfl_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=gen)).to(torch.bfloat16).to(dev)
    b = (torch.randn(n, k, generator=gen) * 0.1).to(torch.bfloat16).to(dev)
    c = torch.full((m, n), float('nan'), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cfl_gemm_bf16_nt(a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, m, n, k, variant, st), 'cfl_gemm_bf16_nt')
    ref = a.float() @ b.float().t()
    scale = float(ref.abs().max())
    err = float((c.float() - ref).abs().max())
    assert err <= 2 ** -8 * scale + 1e-6, (err, scale)


@pytest.mark.parametrize('m,n,k', [(256, 64, 64), (1000, 128, 256), (50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512),
                                   (777, 256, 128), (33, 64, 1024), (4099, 512, 256), (100000, 256, 64), (3000, 128, 512)])
@pytest.mark.parametrize('join', [False, True])
def test_gemm_bf16_nt_b_resident(dev, m, n, k, join):
    """The B-resident streaming kernel (variant 90 / the join entry): the weight tile stays in LDS, every wave streams its own
    32-row tiles of A through registers.  Plain: vs an fp32 matmul of the same bf16 inputs at one bf16 rounding of the output.
    Join (C = (A B^T + add) . mask): vs the same formula in fp32; masked-out elements must be exact zeros.  Ragged M, every
    (K steps, tile width) instantiation, more row ranges than tiles (m = 33)."""
    from creamfl_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=gen)).to(torch.bfloat16).to(dev)
    b = (torch.randn(n, k, generator=gen) * 0.1).to(torch.bfloat16).to(dev)
    c = torch.full((m, n), float('nan'), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ref = a.float() @ b.float().t()
    if not join:
        _lib.check(lib.cfl_gemm_bf16_nt(a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, m, n, k, 90, st), 'cfl_gemm_bf16_nt')
        scale = float(ref.abs().max())
        assert float((c.float() - ref).abs().max()) <= 2 ** -8 * scale + 1e-6
        return
    add = torch.randn(m, n, generator=gen).to(torch.bfloat16).to(dev)
    bits = (torch.rand(m * n, generator=gen) < 0.6).to(dev)
    mask = (bits.view(-1, 8).to(torch.int32) * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(1).to(torch.uint8)
    old = lib.cfl_gemm_bf16_bres_min_m(0)               # every M through the streaming kernel (the product switches at 32768)
    try:
        _lib.check(lib.cfl_gemm_bf16_nt_join(a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), add.data_ptr(), mask.data_ptr(), m, n,
                                             k, st), 'cfl_gemm_bf16_nt_join')
    finally:
        lib.cfl_gemm_bf16_bres_min_m(old)
    keep = bits.view(m, n)
    # the kernel rounds the product to bf16, adds the skip gradient in fp32 and rounds again
    want = (ref.to(torch.bfloat16).float() + add.float()) * keep
    scale = float(want.abs().max())
    assert float((c.float() - want).abs().max()) <= 2 ** -7 * scale + 1e-6
    assert bool((c[~keep] == 0).all())


# ------------------------------------------------------------- A2c / tower glue / KD vs the reference-generated fixtures
import sys as _sys
_sys.path.insert(0, GOLDEN)
from seeded import seeded_state_dict, checksum, resnet_client_template, text_client_template, tower_template  # noqa: E402


def _gradkeys(z, prefix):
    return {k[len(prefix):].replace('__', '.'): z[k] for k in z.files if k.startswith(prefix)}


def _scale_close(got, want, rel, msg='', floor=0.0):
    """|got - want| <= rel * (|want| + max|want|); `floor` = absolute noise level for quantities that are zero in exact
    arithmetic (e.g. the gradient of a bias in front of a train-mode BatchNorm1d: pure round-off on both sides)."""
    want = np.asarray(want, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, dtype=np.float64), want, rtol=rel,
                               atol=max(rel * (np.abs(want).max() + 1e-30), floor), err_msg=msg)


@pytest.mark.parametrize('fname', golden_files('a2c_img_'))
def test_a2c_resnet_client_mirror(dev, fname):
    """creamfl_amd.networks.resnet_client.ResNet (fp32 on the GPU; l2norm on the HIP kernel) vs the reference's
    resnet_client.ResNet.forward: state_dict keys load strictly, both phases, running statistics, weight clamp."""
    from creamfl_amd.networks import resnet_client as rc
    z = _load(fname)
    d, train = int(z['embed_dim']), bool(z['train'])
    sd = seeded_state_dict(resnet_client_template(d), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)
    torch.backends.cudnn.allow_tf32 = False
    model = rc.resnet10_client(embed_dim=d, num_class=10, is_train=True, scale=128, phase='none')
    model.load_state_dict(sd, strict=True)                 # key-name parity with the reference's module
    model = model.to(dev).train(train)
    x = torch.from_numpy(z['x']).to(dev)
    named = dict(model.named_parameters())
    model.phase = 'extract_conv_feature'                   # ClientTrainer.py:372-375 switches modes by mutation
    feat = model(x)
    _scale_close(feat, z['feat'], 1e-4)
    (feat * torch.from_numpy(z['gy']).to(dev)).sum().backward()
    for k, g in _gradkeys(z, 'feat__g_').items():
        _scale_close(named[k].grad, g, 1e-3, k)
    if train:
        _scale_close(model.state_dict()['bn1.running_mean'], z['bn1_running_mean_after_feat'], 1e-4)
    model.phase = 'none'
    model.zero_grad()
    x1, x2, w, w2 = model(x)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (w, 'w'), (w2, 'w2')):
        _scale_close(got, z[key], 1e-4, key)
    np.testing.assert_array_equal(model.class_fc_2.weight.detach().cpu().numpy(), z['class_fc_2_weight_after'])
    np.testing.assert_array_equal(model.class_fc_22.weight.detach().cpu().numpy(), z['class_fc_22_weight_after'])
    ((x1 * torch.from_numpy(z['g1']).to(dev)).sum() + (x2 * torch.from_numpy(z['g2']).to(dev)).sum() + 0.1 * (w ** 2).sum()).backward()
    for k, g in _gradkeys(z, 'cls__g_').items():
        _scale_close(named[k].grad, g, 1e-3, k)


@pytest.mark.parametrize('fname', golden_files('a2c_txt_'))
def test_a2c_text_client_mirror(dev, fname):
    """creamfl_amd.networks.language_model.EncoderText (HIP PIE head + l2norm) vs the reference's EncoderText.forward."""
    from creamfl_amd.networks.language_model import EncoderText
    z = _load(fname)
    d = int(z['embed_dim'])
    sd = seeded_state_dict(text_client_template(d, int(z['vocab'])), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)
    model = EncoderText(wemb_type=None, word_dim=300, embed_dim=d, num_class=4, scale=128, vocab_size=int(z['vocab']))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    named = dict(model.named_parameters())
    x, lengths = torch.from_numpy(z['x']).to(dev), torch.from_numpy(z['lengths'])
    model.is_train = False
    feat = model(x, lengths)
    _scale_close(feat, z['feat'], 1e-4)
    (feat * torch.from_numpy(z['gy']).to(dev)).sum().backward()
    for k, g in _gradkeys(z, 'feat__g_').items():
        _scale_close(named[k].grad, g, 1e-3, k)
    model.is_train = True
    model.zero_grad()
    x1, x2, w, w2 = model(x, lengths)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (w, 'w'), (w2, 'w2')):
        _scale_close(got, z[key], 1e-4, key)
    np.testing.assert_array_equal(model.class_fc.weight.detach().cpu().numpy(), z['class_fc_weight_after'])
    ((x1 * torch.from_numpy(z['g1']).to(dev)).sum() + (x2 * torch.from_numpy(z['g2']).to(dev)).sum() + 0.1 * (w ** 2).sum()).backward()
    for k, g in _gradkeys(z, 'cls__g_').items():
        _scale_close(named[k].grad, g, 1e-3, k)


class _IdentityTrunk(torch.nn.Module):
    """Stands where the ResNet trunk is: the fixtures hold the trunk's OUTPUT map (as the reference generator did)."""

    def __init__(self, out_dim):
        super().__init__()
        self.out_dim = out_dim

    def features(self, x):
        return x


@pytest.mark.parametrize('fname', golden_files('tower_'))
@pytest.mark.parametrize('channels_last', [False, True])
def test_a2_pcme_tower_glue_mirror(dev, fname, channels_last):
    """creamfl_amd PCME.forward / EncoderImage.forward / EncoderText.forward (fused HIP PIE head, mean-pool + fc,
    head_proj, l2norm) vs the reference's own PCME.forward run on the same trunk output; 10-key dict, both memory
    formats of the trunk output (the channels_last one is the zero-copy [N, 49, Cd] view)."""
    from creamfl_amd.networks.models.pcme import PCME
    from creamfl_amd.networks.models.image_encoder import EncoderImage
    from creamfl_amd.networks.models.caption_encoder import EncoderText
    from creamfl_amd.networks.models.pie_model import PIENet
    from creamfl_amd.utils.config import Config as AttrDict
    nn = torch.nn
    z = _load(fname)
    cd, d, mlp = int(z['cd']), int(z['embed_dim']), bool(z['mlp_local'])
    sd = seeded_state_dict(tower_template(cd, d, mlp), int(z['seed']))
    np.testing.assert_allclose(checksum(sd), float(z['wsum']), rtol=1e-12)
    cfg = AttrDict(embed_dim=d, wemb_type=None, word_dim=300, cache_dir=None, not_bert=True, n_samples_inference=7,
                   cnn_type='resnet18')
    img = EncoderImage.__new__(EncoderImage)
    nn.Module.__init__(img)
    img.cnn, img.cnn_dim = _IdentityTrunk(cd), cd
    img.avgpool = nn.AdaptiveAvgPool2d((1, 1))
    img.fc = nn.Linear(cd, d)
    img.pie_net = PIENet(1, cd, d, cd // 2)
    img.mlp_local = mlp
    if mlp:
        img.head_proj = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Linear(512, 512))
    model = PCME.__new__(PCME)
    nn.Module.__init__(model)
    model.config, model.embed_dim, model.n_embeddings = cfg, d, 7
    model.img_enc = img
    model.txt_enc = EncoderText({str(i): i for i in range(60)}, cfg, mlp)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    fmap = torch.from_numpy(z['fmap']).to(dev)
    if channels_last:
        fmap = fmap.contiguous(memory_format=torch.channels_last)
    fmap.requires_grad_(True)
    out = model(fmap, torch.from_numpy(z['sentences']).to(dev), None, torch.from_numpy(z['lengths']).to(dev))
    assert list(out.keys()) == [str(k) for k in z['keys']]
    assert [k for k in out if out[k] is None] == [str(k) for k in z['none_keys']]
    _scale_close(out['image_features'], z['image_features'], 1e-4)
    _scale_close(out['caption_features'], z['caption_features'], 1e-4)
    ((out['image_features'] * torch.from_numpy(z['gi']).to(dev)).sum()
     + (out['caption_features'] * torch.from_numpy(z['gc']).to(dev)).sum()).backward()
    _scale_close(fmap.grad, z['dfmap'], 1e-3)
    named = dict(model.named_parameters())
    for k, g in _gradkeys(z, 'g_').items():
        _scale_close(named[k].grad, g, 1e-3, k, floor=1e-6)


@pytest.mark.parametrize('fname', golden_files('kd_'))
def test_f1_kd_terms_golden(dev, fname):
    """MMFL.kd_terms (fused gather + MSE kernel per term) vs the literal MMFL.py:346-378 sequence: every combination
    of client types incl. the doubled image term, [B, 7, D] outputs summed over axis 1."""
    from types import SimpleNamespace
    from creamfl_amd.algorithms.MMFL import MMFL
    z = _load(fname)
    me = SimpleNamespace(args=SimpleNamespace(num_img_clients=int(z['num_img_clients']), num_txt_clients=int(z['num_txt_clients']),
                                              num_mm_clients=int(z['num_mm_clients']), kd_weight=float(z['kd_weight'])),
                         img_vec=torch.from_numpy(z['img_vec']).to(dev), txt_vec=torch.from_numpy(z['txt_vec']).to(dev))
    oi = torch.from_numpy(z['out_img']).to(dev).requires_grad_(True)
    ot = torch.from_numpy(z['out_txt']).to(dev).requires_grad_(True)
    dd = {int(b): a for a, b in enumerate(z['distill_index'])}
    d_idx = torch.as_tensor([dd[int(i)] for i in z['index']], device=dev)
    loss = MMFL.kd_terms(me, {'image_features': oi, 'caption_features': ot}, d_idx)
    loss.backward()
    _close(loss.item(), float(z['loss']), 1e-5, 0)
    _scale_close(oi.grad if oi.grad is not None else torch.zeros_like(oi), z['d_out_img'], 1e-5)
    _scale_close(ot.grad if ot.grad is not None else torch.zeros_like(ot), z['d_out_txt'], 1e-5)


# ------------------------------------------------------------------ the dense kernels on the 3 x bf16-split and the exact fp32 tile GEMM
@pytest.mark.parametrize('exact', [0, 1])
def test_dense_kernels_in_both_precisions(dev, exact):
    """The pair loss (A1), the two-pass bank kernels (A3 at D > 256) and con_w (A5) against their goldens / fp64 closed forms
    with the dense core on the 3 x bf16-split MFMA (default) and on the exact fp32 MFMA (cfl_set_exact_gemm)."""
    from creamfl_amd import _lib, ops
    lib = _lib.load()
    old = lib.cfl_get_exact_gemm()
    lib.cfl_set_exact_gemm(exact)
    try:
        z = _load('a1_n128_d256_a15_b15.npz')
        loss, stats, dI, dT, da, db = _run_pair(dev, torch.from_numpy(z['I']), torch.from_numpy(z['T']), float(z['a']), float(z['b']))
        cf = oracle.pair_loss_closed_form(torch.from_numpy(z['I']), torch.from_numpy(z['T']), float(z['a']), float(z['b']))
        g = oracle.pair_loss_grads_closed_form(torch.from_numpy(z['I']), torch.from_numpy(z['T']), float(z['a']), float(z['b']))
        _close(loss, float(cf['loss']), 3e-5, 1e-6)
        sc = float(np.abs(np.asarray(g['dI'])).max())
        _close(dI, np.asarray(g['dI']), 1e-4, 2e-4 * sc)
        # two-pass bank kernels (D = 320 > 256 keeps the single-pass kernel out of the way)
        gen = torch.Generator().manual_seed(7)
        G = _unit(gen, 3000, 320)
        idx = torch.randint(0, 3000, (96,), generator=gen)
        f = torch.nn.functional.normalize(G[idx] + 0.7 * _unit(gen, 96, 320), dim=-1)
        fg = f.to(dev).requires_grad_(True)
        l2, lse, pos = ops.inter_contrast(fg, G.to(dev), idx.tolist(), 0.5)
        l2.backward()
        c2 = oracle.client_contrast_grads_closed_form(f, G, G, idx.tolist(), f)
        _close(l2.item(), c2['loss_inter'].item(), 2e-5, 1e-5)
        dref = c2['d_inter'].numpy()
        _close(fg.grad.cpu().numpy(), dref, 1e-4, 3e-5 * np.abs(dref).max())
        # con_w
        z5 = _load('a5_m600_d128_c4.npz')
        vecs = [torch.from_numpy(v).to(dev) for v in z5['vecs']]
        Gd = torch.from_numpy(z5['g_other']).to(dev)
        lp = torch.stack([ops.conw_logprob(v, Gd) for v in vecs], 0)
        agg, w = ops.conw_combine(vecs, lp, return_weights=True)
        _close(lp.cpu().numpy(), z5['logprob'], 1e-5, 1e-5)
        _close(w.cpu().numpy(), z5['weights'], 1e-4, 1e-6)
    finally:
        lib.cfl_set_exact_gemm(old)


def test_kernel_profiler_counts_and_times_launches(dev):
    """The per-kernel profiler behind the C ABI (what bench.py's `roofline` and tools/kernel_bench.py read): a profiled launch goes
    through hipExtLaunchKernelGGL with a start and a stop event.  Launch counts are exact, `prof_select` restricts them to one
    kernel, the duration of a 205 MB GEMM is in the range HBM allows (not the ~zero of two back-to-back markers, not a
    host-side interval), and results are identical with the profiler on and off."""
    from creamfl_amd import _lib, ops
    m, n, k = 50176, 1024, 256
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(dev)
    b = (torch.randn(n, k, generator=gen) * 0.1).to(torch.bfloat16).to(dev)
    ref = ops.gemm_bf16_nt(a, b)
    x = torch.randn(4096, 64, generator=gen).to(dev)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_select(None)
    _lib.prof_enable(True)
    try:
        outs = [ops.gemm_bf16_nt(a, b) for _ in range(5)]
        ops.l2_normalize(x)
        torch.cuda.synchronize()
        q = _lib.prof_query()
        assert q['cfl_gemm_bf16_kernel'][0] == 5
        us = q['cfl_gemm_bf16_kernel'][1] / 5 * 1e3
        assert 15.0 < us < 400.0, us                       # 128 MB of traffic: 16 us at the HBM peak
        assert sum(v[0] for v in q.values()) >= 6
        _lib.prof_reset()
        _lib.prof_select('cfl_gemm_bf16_kernel')
        ops.gemm_bf16_nt(a, b)
        ops.l2_normalize(x)
        torch.cuda.synchronize()
        q = _lib.prof_query()
        assert list(q) == ['cfl_gemm_bf16_kernel'] and q['cfl_gemm_bf16_kernel'][0] == 1
    finally:
        _lib.prof_enable(False)
        _lib.prof_select(None)
        _lib.prof_reset()
    for o in outs:
        assert torch.equal(o, ref)


# ------------------------------------------------------------------------- boundary (b): the DOCUMENTED binding
def test_integration_md_binding_stubs_run_against_the_oracle(dev):
    """INTEGRATION.md section B is the reference-side ctypes binding a maintainer would add next to src/criterions/probemb.py and
    src/algorithms/ClientTrainer.py.  Both stubs are extracted from the document and executed as printed (conftest.
    integration_stubs): the pair loss (probemb.py:221-256) against the fp64 closed forms, the client contrast block
    (ClientTrainer.py:386-419, both --loss_scale settings) against the oracle -- the tolerances of the rest of this file."""
    from conftest import integration_stubs
    ns = integration_stubs()
    gen = torch.Generator().manual_seed(77)
    N, D = 96, 128
    I = _unit(gen, N, D)
    T = torch.nn.functional.normalize(I + 0.5 * _unit(gen, N, D), dim=-1)
    Ig, Tg = I.to(dev).requires_grad_(True), T.to(dev).requires_grad_(True)
    a = torch.tensor([15.0], device=dev, requires_grad=True)
    b = torch.tensor([14.0], device=dev, requires_grad=True)
    loss = ns['PairLoss'].apply(Ig, Tg, a, b)
    loss.backward()
    cf = oracle.pair_loss_closed_form(I, T, 15.0, 14.0)
    gr = oracle.pair_loss_grads_closed_form(I, T, 15.0, 14.0)
    _close(loss.item(), cf['loss'].item(), 2e-5, 0, 'stub pair loss')
    for got, want, nm in ((Ig.grad, gr['dI'], 'dI'), (Tg.grad, gr['dT'], 'dT')):
        _close(got.cpu().numpy(), want.numpy(), 1e-4, 1e-4 * float(want.abs().max()), 'stub ' + nm)
    _close(a.grad.item(), gr['da'].item(), 1e-4, 1e-3, 'stub da')
    _close(b.grad.item(), gr['db'].item(), 1e-4, 1e-3, 'stub db')
    # the client contrast block
    B, M = 48, 3000
    G, Gs = _unit(gen, M, D), _unit(gen, M, D)
    idx = torch.randperm(M, generator=gen)[:B]
    f, fo = _unit(gen, B, D), _unit(gen, B, D)
    for loss_scale in (False, True):
        fg = f.to(dev).requires_grad_(True)
        l = ns['ClientContrast'].apply(fg, Gs.to(dev), G.to(dev), idx.to(dev), fo.to(dev), 0.5, loss_scale)
        l.backward()
        f64 = f.double().requires_grad_(True)
        lo, _, _ = oracle.client_contrast_loss(f64, Gs.double(), G.double(), idx.tolist(), fo.double(), 0.5, loss_scale)
        lo.backward()
        _close(l.item(), lo.item(), 1e-4, 0, 'stub client contrast loss_scale=%s' % loss_scale)
        _close(fg.grad.cpu().numpy(), f64.grad.numpy(), 1e-3, 1e-3 * float(f64.grad.abs().max()), 'stub df')
