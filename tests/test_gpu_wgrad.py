"""GPU: the hand-written 3 x 3 weight gradient (csrc/wgrad3x3.hip, round 6) against the convolution backward evaluated in fp32 / fp64 on
the same bf16-valued inputs (torchvision Bottleneck.conv2 inside src/networks/models/image_encoder.py:27-36; the reference leaves the
operation to cuDNN, so the check is against the definition: dW[co, ci, kh, kw] = sum_{n,h,w} dY[n, co, h, w] X[n, ci, h+kh-1, w+kw-1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_dw(x, dy, ksize=3):
    """fp64 on the CPU for small problems, fp32 aten on the GPU for the large ones (bf16-valued inputs either way)."""
    big = x.numel() > (1 << 22)
    xf, df = (x.float(), dy.float()) if big else (x.double().cpu(), dy.double().cpu())
    w0 = torch.zeros(dy.shape[1], x.shape[1], ksize, ksize, dtype=xf.dtype, device=xf.device)
    return torch.ops.aten.convolution_backward(df, xf, w0, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


@pytest.mark.parametrize('n,h,ci,co', [(8, 14, 64, 128), (37, 14, 128, 128), (3, 14, 256, 256), (40, 7, 128, 256), (9, 7, 512, 512),
                                       (256, 14, 256, 256), (5, 28, 128, 128), (19, 28, 64, 256), (3, 56, 64, 64), (11, 56, 128, 64),
                                       (64, 28, 128, 128), (32, 56, 64, 64)])
def test_conv3x3_wgrad_matches_the_definition(n, h, ci, co):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(100 * n + ci + h)
    cl = torch.channels_last
    # non-symmetric data with a non-zero mean (a padding / shift mistake moves a mean-carrying sum by much more than the tolerance)
    x = (torch.randn(n, ci, h, h, generator=g) * 0.8 + 0.25).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    dy = (torch.randn(n, co, h, h, generator=g) * 0.5 - 0.1).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    w = torch.zeros(co, ci, 3, 3, dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
    before = ops.WGRAD3_TAKEN[0]
    dw = ops.conv3x3_wgrad(dy, x, w)
    assert dw is not None and ops.WGRAD3_TAKEN[0] == before + 1
    assert dw.shape == w.shape and dw.dtype == torch.bfloat16 and dw.is_contiguous(memory_format=cl)
    ref = _ref_dw(x, dy).float().cpu()
    got = dw.float().cpu()
    scale = float(ref.abs().max())
    # fp32 accumulation in another order + ONE bf16 rounding of the result (2^-9 relative); the reference of the big case is itself fp32
    err = (got - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) <= 2e-3 * scale, (float(err.max()), scale)
    # every tap is its own sum: compare tap by tap so that a swapped / shifted tap cannot hide in the maximum over all of them
    for kh in range(3):
        for kw in range(3):
            e = float((got[:, :, kh, kw] - ref[:, :, kh, kw]).abs().max())
            assert e <= 2.0 ** -7 * float(ref[:, :, kh, kw].abs().max()) + 2e-3 * scale, (kh, kw, e)
    # deterministic: fixed-order split-K reduce
    dw2 = ops.conv3x3_wgrad(dy, x, w)
    assert torch.equal(dw, dw2)


def test_conv3x3_wgrad_split_override_and_refusals():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device('cuda:0')
    cl = torch.channels_last
    g = torch.Generator().manual_seed(5)
    x = torch.randn(24, 256, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    dy = torch.randn(24, 256, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    w = torch.zeros(256, 256, 3, 3, dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
    ref = _ref_dw(x, dy).float().cpu()
    old = lib.cfl_conv3x3_wgrad_splits(-1)
    try:
        for splits in (8, 16, 0):
            lib.cfl_conv3x3_wgrad_splits(splits)
            got = ops.conv3x3_wgrad(dy, x, w).float().cpu()
            assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    finally:
        lib.cfl_conv3x3_wgrad_splits(old)
    # shapes / layouts the kernel does not take go to the library (None here)
    assert lib.cfl_conv3x3_wgrad_supported(8, 28, 28, 128, 128) == 1 and lib.cfl_conv3x3_wgrad_supported(8, 14, 14, 96, 128) == 0
    assert lib.cfl_conv3x3_wgrad_supported(8, 14, 14, 64, 64) == 0 and lib.cfl_conv3x3_wgrad_supported(8, 14, 14, 64, 128) == 1
    assert lib.cfl_conv3x3_wgrad_supported(8, 56, 56, 64, 64) == 1 and lib.cfl_conv3x3_wgrad_supported(8, 56, 56, 64, 128) == 0
    assert lib.cfl_conv3x3_wgrad_supported(8, 112, 112, 64, 64) == 0 and lib.cfl_conv3x3_wgrad_supported(8, 14, 7, 64, 128) == 0
    assert ops.conv3x3_wgrad(dy.contiguous(), x, w) is None                              # NCHW gradient
    assert ops.conv3x3_wgrad(dy.float(), x.float(), w.float()) is None                  # fp32 (the clients' encoders)
    x9 = torch.zeros(2, 128, 9, 9, dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
    w9 = torch.zeros(128, 128, 3, 3, dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
    assert ops.conv3x3_wgrad(x9, x9, w9) is None


def test_conv_split_backward_uses_the_kernel_and_matches_the_library():
    """Through the product's autograd node (ops.conv_split, side-stream weight gradient): the weight gradient of a layer3-shaped
    convolution comes from csrc/wgrad3x3.hip and equals MIOpen's to bf16 rounding."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops, streams
    dev = torch.device('cuda:0')
    cl = torch.channels_last
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16, 256, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.05).to(torch.bfloat16).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    gy = torch.randn(16, 256, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    grads = {}
    for on in (True, False):
        ops.WGRAD3[0] = on
        try:
            x.grad = w.grad = None
            before = ops.WGRAD3_TAKEN[0]
            y = ops.conv_split(x, w, stride=1, padding=1)
            y.backward(gy)
            streams.join_into_current(dev)
            torch.cuda.synchronize()
            assert (ops.WGRAD3_TAKEN[0] - before) == (1 if on else 0)
            grads[on] = (w.grad.float().cpu().clone(), x.grad.float().cpu().clone())
        finally:
            ops.WGRAD3[0] = True
    sc = float(grads[False][0].abs().max())
    np.testing.assert_allclose(grads[True][0].numpy(), grads[False][0].numpy(), rtol=2.0 ** -6, atol=4e-3 * sc)
    assert torch.equal(grads[True][1], grads[False][1])                 # the data gradient is untouched


@pytest.mark.parametrize('n,h,ci,co', [(8, 14, 1024, 256), (8, 14, 256, 1024), (5, 7, 2048, 512), (5, 7, 512, 2048), (3, 7, 1024, 512),
                                       (4, 28, 512, 128), (4, 28, 128, 512), (2, 56, 256, 128), (2, 56, 256, 64), (2, 56, 64, 256),
                                       (2, 56, 64, 64), (3, 7, 256, 256), (1, 5, 256, 64), (256, 14, 1024, 256), (64, 28, 128, 512)])
def test_conv1x1_wgrad_matches_the_definition(n, h, ci, co, monkeypatch):
    """csrc/wgrad1x1.hip: dW[co, ci] = sum_m dY[m, co] X[m, ci] against fp64 (small) / fp32 (large) on the same bf16-valued inputs; row
    counts that are no multiple of the 16-row stage or of the split count; deterministic.  (Every map size the kernel takes: the
    step's own gate, ops.WGRAD1_MAX_HW = 28, leaves the 56 x 56 layers to the library on an in-step A/B, not for correctness.)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import ops
    assert ops.WGRAD1_MAX_HW[0] == 28
    monkeypatch.setattr(ops, 'WGRAD1_MAX_HW', [0])
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7 * n + ci + co + h)
    cl = torch.channels_last
    x = (torch.randn(n, ci, h, h, generator=g) * 0.8 + 0.25).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    dy = (torch.randn(n, co, h, h, generator=g) * 0.5 - 0.1).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    w = torch.zeros(co, ci, 1, 1, dtype=torch.bfloat16, device=dev)
    before = ops.WGRAD1_TAKEN[0]
    dw = ops.conv1x1_wgrad(dy, x, w)
    assert dw is not None and ops.WGRAD1_TAKEN[0] == before + 1 and dw.shape == w.shape and dw.dtype == torch.bfloat16
    M = n * h * h
    xm, dm = x.permute(0, 2, 3, 1).reshape(M, ci), dy.permute(0, 2, 3, 1).reshape(M, co)
    if M * (ci + co) > (1 << 23):
        ref = (dm.float().t() @ xm.float()).cpu()
    else:
        ref = (dm.double().cpu().t() @ xm.double().cpu()).float()
    got = dw.float().cpu().view(co, ci)
    scale = float(ref.abs().max())
    err = (got - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) <= 2e-3 * scale, (float(err.max()), scale)
    assert torch.equal(dw, ops.conv1x1_wgrad(dy, x, w))
    # the channels_last weight of the trunk (same memory for a 1 x 1 kernel) is taken as well; other layouts / dtypes / shapes are not
    assert ops.conv1x1_wgrad(dy, x, w.contiguous(memory_format=cl)) is not None
    assert ops.conv1x1_wgrad(dy.float(), x.float(), w.float()) is None and ops.conv1x1_wgrad(dy.contiguous(), x, w) is None


def test_conv1x1_wgrad_through_conv_split_matches_the_library():
    """The product's autograd node with a bottleneck's conv3 shape: weight gradient from csrc/wgrad1x1.hip, equal to the library's to
    bf16 rounding; the data gradient (the hand-written GEMM either way) bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib, ops, streams
    dev = torch.device('cuda:0')
    cl = torch.channels_last
    g = torch.Generator().manual_seed(12)
    x = torch.randn(16, 256, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    w = (torch.randn(1024, 256, 1, 1, generator=g) * 0.05).to(torch.bfloat16).to(dev).contiguous(memory_format=cl).requires_grad_(True)
    gy = torch.randn(16, 1024, 14, 14, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=cl)
    grads = {}
    for on in (True, False):
        ops.WGRAD1[0] = on
        try:
            x.grad = w.grad = None
            before = ops.WGRAD1_TAKEN[0]
            y = ops.conv_split(x, w, stride=1, padding=0)
            y.backward(gy)
            streams.join_into_current(dev)
            torch.cuda.synchronize()
            assert (ops.WGRAD1_TAKEN[0] - before) == (1 if on else 0)
            grads[on] = (w.grad.float().cpu().clone(), x.grad.float().cpu().clone())
        finally:
            ops.WGRAD1[0] = True
    sc = float(grads[False][0].abs().max())
    np.testing.assert_allclose(grads[True][0].numpy(), grads[False][0].numpy(), rtol=2.0 ** -6, atol=4e-3 * sc)
    assert torch.equal(grads[True][1], grads[False][1])
    lib = _lib.load()
    old = lib.cfl_conv1x1_wgrad_workgroups(256)
    try:
        a = ops.conv1x1_wgrad(gy, x.detach(), w.detach()).float().cpu()
    finally:
        lib.cfl_conv1x1_wgrad_workgroups(old)
    np.testing.assert_allclose(a.numpy(), grads[False][0].numpy(), rtol=2.0 ** -6, atol=4e-3 * sc)
