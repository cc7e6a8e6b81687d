"""GPU: csrc/conv3x3_x3.hip -- the 3 x 3 / stride 1 / padding 1 convolution of fp32 channels_last tensors as 3 x bf16-split products
on the bf16 matrix pipe (the clients' ResNet-18 BasicBlocks, src/networks/resnet_client.py:33-66,102-201; fp32 on cuDNN in the
reference) -- against the definition in fp64 and against the library's fp32 kernels it replaces."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from creamfl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _case(n, h, w, ci, co, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    return x, wt


@pytest.mark.parametrize('n,h,w,ci,co,variant', [(2, 7, 7, 64, 64, 0), (3, 14, 14, 64, 128, 22), (1, 28, 28, 128, 128, 42), (5, 9, 13, 32, 64, 21),
                                                 (4, 56, 56, 64, 64, 41), (1, 1, 1, 32, 64, 0), (2, 5, 3, 96, 192, 0), (16, 7, 7, 512, 512, 0),
                                                 (3, 14, 14, 64, 128, 122), (2, 28, 28, 128, 128, 142), (5, 9, 13, 32, 64, 121),
                                                 (3, 56, 56, 64, 64, 141), (1, 1, 1, 32, 64, 121), (2, 5, 3, 96, 192, 121),
                                                 (16, 7, 7, 512, 512, 122), (1, 63, 63, 32, 64, 121), (7, 7, 7, 64, 64, 141),
                                                 (3, 14, 14, 64, 128, 222), (2, 28, 28, 128, 128, 242), (5, 9, 13, 32, 64, 221),
                                                 (3, 56, 56, 64, 64, 241), (1, 1, 1, 32, 64, 211), (2, 5, 3, 96, 192, 221),
                                                 (16, 7, 7, 512, 512, 212), (1, 63, 63, 32, 64, 211), (7, 7, 7, 64, 64, 241),
                                                 (9, 7, 7, 128, 256, 222), (2, 64, 64, 32, 64, 0), (4, 28, 28, 64, 128, 212),
                                                 (3, 14, 14, 64, 128, 522), (2, 28, 28, 128, 128, 542), (5, 9, 13, 32, 64, 521),
                                                 (16, 7, 7, 512, 512, 522), (3, 56, 56, 64, 64, 521), (1, 1, 1, 32, 128, 522),
                                                 (3, 14, 14, 64, 128, 722), (2, 28, 28, 128, 128, 742), (5, 9, 13, 32, 64, 721),
                                                 (16, 7, 7, 512, 512, 722), (3, 56, 56, 64, 64, 721), (1, 1, 1, 32, 128, 722),
                                                 (2, 5, 3, 96, 192, 721), (37, 14, 14, 256, 256, 722),
                                                 (3, 14, 14, 64, 128, 822), (2, 28, 28, 128, 128, 842), (5, 9, 13, 32, 64, 821),
                                                 (16, 7, 7, 512, 512, 822), (3, 56, 56, 64, 64, 821), (1, 1, 1, 32, 128, 822)])
def test_x3conv_forward_matches_the_definition(dev, n, h, w, ci, co, variant):
    """y = conv2d(x, w, 1, 1) against fp64 on the same fp32 inputs: 1e-5 of the output's scale (a 3 x bf16-split product is ~1e-6
    relative; the library's fp32 Winograd kernels are ~1e-3 of scale); ragged position counts, every tile variant, single pixels."""
    from creamfl_amd import ops
    x, wt = _case(n, h, w, ci, co, n + h + ci)
    xd, wd = x.to(dev).contiguous(memory_format=CL), wt.to(dev).contiguous(memory_format=CL)
    assert ops.conv3x3_x3_supported(xd, wd, 1, 1)
    y = ops.conv3x3_x3_forward(xd, wd, variant)
    assert y.shape == (n, co, h, w) and y.is_contiguous(memory_format=CL)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    sc = float(ref.abs().max())
    assert float((y.cpu().double() - ref).abs().max()) <= 1e-5 * sc
    assert torch.equal(y, ops.conv3x3_x3_forward(xd, wd, variant))                 # deterministic
    assert not ops.conv3x3_x3_supported(xd, wd, 3, 1)
    assert ops.conv3x3_x3_supported(xd, wd, 2, 1) == (h % 2 == 0 and w % 2 == 0 and w <= 62)      # (forward only: round 6)
    if h * w > 1:                                                                  # (a 1 x 1 map is contiguous in both formats)
        assert not ops.conv3x3_x3_supported(xd.contiguous(), wd, 1, 1)
    assert not ops.conv3x3_x3_supported(xd.bfloat16(), wd.bfloat16(), 1, 1)


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 56, 56, 64, 128), (3, 28, 28, 128, 256), (5, 14, 14, 256, 512), (1, 2, 2, 32, 64), (4, 6, 10, 32, 64),
                                         (7, 62, 62, 32, 64), (37, 14, 14, 64, 64), (128, 8, 4, 32, 192), (3, 62, 2, 64, 128)])
def test_x3conv_stride2_forward_matches_the_definition(dev, n, h, w, ci, co):
    """Round 6: y = conv2d(x, w, stride 2, padding 1) on the same kernel (a tile of output positions still reads one contiguous range of
    input positions; only a lane's nine row addresses change) against fp64: the three down-sampling convolutions of ResNet-18 and the
    corners -- a 2 x 2 map, the widest map, tiles that cut rows and images, row / image wraps inside a tile."""
    from creamfl_amd import ops
    x, wt = _case(n, h, w, ci, co, 3 * n + h + ci)
    x = x * 0.8 + 0.25                                                             # (a mean: a padding mistake moves sums far beyond the tolerance)
    xd, wd = x.to(dev).contiguous(memory_format=CL), wt.to(dev).contiguous(memory_format=CL)
    assert ops.conv3x3_x3_supported(xd, wd, 2, 1)
    y = ops.conv3x3_x3_forward(xd, wd, stride=2)
    assert y.shape == (n, co, h // 2, w // 2) and y.is_contiguous(memory_format=CL)
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    sc = float(ref.abs().max())
    assert float((y.cpu().double() - ref).abs().max()) <= 1e-5 * sc
    assert torch.equal(y, ops.conv3x3_x3_forward(xd, wd, stride=2))


@pytest.mark.parametrize('n,h,ci,co', [(2, 7, 64, 64), (9, 7, 512, 512), (5, 14, 64, 128), (16, 14, 256, 256), (3, 28, 128, 128),
                                       (7, 28, 64, 192), (2, 56, 64, 64), (3, 56, 128, 64), (1, 14, 64, 64), (37, 7, 128, 64)])
def test_x3_wgrad_matches_the_definition(dev, n, h, ci, co):
    """dW[co, ci, kh, kw] = sum_{n,h,w} dY[n, co, h, w] X[n, ci, h+kh-1, w+kw-1] (csrc/wgrad3x3_x3.hip) against fp64 on the same fp32
    inputs: 1e-5 of the gradient's scale, tap by tap (a swapped / shifted tap cannot hide in the maximum over all of them; the data
    carry a mean so that a padding mistake moves a sum by far more than the tolerance); row ranges that cut images, ragged counts."""
    from creamfl_amd import ops
    g = torch.Generator().manual_seed(100 * n + ci + h)
    x = (torch.randn(n, ci, h, h, generator=g) * 0.8 + 0.25)
    dy = (torch.randn(n, co, h, h, generator=g) * 0.5 - 0.1)
    xd, dyd = x.to(dev).contiguous(memory_format=CL), dy.to(dev).contiguous(memory_format=CL)
    w = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=CL)
    before = ops.X3WGRAD_TAKEN[0]
    dw = ops.conv3x3_x3_wgrad(dyd, xd, w)
    assert dw is not None and ops.X3WGRAD_TAKEN[0] == before + 1
    assert dw.shape == w.shape and dw.dtype == torch.float32 and dw.is_contiguous(memory_format=CL)
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.cpu().double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    got = dw.cpu().double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    for kh in range(3):
        for kw in range(3):
            e = float((got[:, :, kh, kw] - ref[:, :, kh, kw]).abs().max())
            assert e <= 1e-5 * scale, (kh, kw, e, scale)
    assert torch.equal(dw, ops.conv3x3_x3_wgrad(dyd, xd, w))                      # deterministic: fixed-order split-K reduce
    lib = ops._lib.load()
    for splits in (8, 24, 64):                                                    # any number of row ranges, the same sums
        old = lib.cfl_conv3x3_x3_wgrad_splits(splits)
        try:
            dws = ops.conv3x3_x3_wgrad(dyd, xd, w)
        finally:
            lib.cfl_conv3x3_x3_wgrad_splits(old)
        assert float((dws.cpu().double() - ref).abs().max()) <= 1e-5 * scale, splits


def test_x3_wgrad_refusals(dev):
    from creamfl_amd import ops
    x = torch.randn(2, 64, 14, 14, device=dev).contiguous(memory_format=CL)
    dy = torch.randn(2, 64, 14, 14, device=dev).contiguous(memory_format=CL)
    w = torch.zeros(64, 64, 3, 3, device=dev).contiguous(memory_format=CL)
    assert ops.conv3x3_x3_wgrad(dy, x, w) is not None
    assert ops.conv3x3_x3_wgrad(dy.bfloat16(), x.bfloat16(), w.bfloat16()) is None         # the bf16 trunk has its own kernel
    assert ops.conv3x3_x3_wgrad(dy.contiguous(), x.contiguous(), w) is None                 # NCHW
    assert ops.conv3x3_x3_wgrad(dy[:, :, :13, :13].contiguous(memory_format=CL), x[:, :, :13, :13].contiguous(memory_format=CL), w) is None
    x32 = torch.randn(2, 32, 14, 14, device=dev).contiguous(memory_format=CL)
    assert ops.conv3x3_x3_wgrad(dy, x32, torch.zeros(64, 32, 3, 3, device=dev).contiguous(memory_format=CL)) is None


@pytest.mark.parametrize('n,hw,ci,co', [(2, 14, 64, 128), (1, 7, 256, 64), (3, 28, 128, 128)])
def test_x3conv_data_gradient_from_the_rotated_image(dev, n, hw, ci, co):
    """dX of y = conv2d(x, w, 1, 1): the forward kernel on the image of the rotated, transposed weight, built in one pass from w
    (cfl_conv3x3_x3_wimage_rot) -- against fp64 autograd, and equal to the two-step form (rotate, then image); the image of a frozen
    weight is kept across calls and rebuilt when the weight changes in place."""
    from creamfl_amd import ops
    _, wt = _case(n, hw, hw, ci, co, 7)
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(n, co, hw, hw, generator=g)
    wd = wt.to(dev).contiguous(memory_format=CL)
    dyd = dy.to(dev).contiguous(memory_format=CL)
    dx = ops.conv3x3_x3_forward(dyd, wd, rotated=True)
    assert dx.shape == (n, ci, hw, hw)
    x64 = torch.zeros(n, ci, hw, hw, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, wt.double(), None, 1, 1).backward(dy.double())
    sc = float(x64.grad.abs().max())
    assert float((dx.cpu().double() - x64.grad).abs().max()) <= 1e-5 * sc
    assert torch.equal(dx, ops.conv3x3_x3_forward(dyd, ops.conv3x3_x3_rotated(wd)))
    # frozen weight: cached image; an in-place change of the weight must not be served from it
    n_img = len(ops._X3_IMAGES)
    assert torch.equal(dx, ops.conv3x3_x3_forward(dyd, wd, rotated=True)) and len(ops._X3_IMAGES) == n_img
    wd.mul_(2.0)
    assert float((ops.conv3x3_x3_forward(dyd, wd, rotated=True) - 2.0 * dx).abs().max()) <= 1e-6 * float(dx.abs().max())
    del wd
    import gc
    gc.collect()
    assert len(ops._X3_IMAGES) <= n_img - 1                                       # the entry died with its tensor


def test_x3conv_rotated_weight_is_exact(dev):
    from creamfl_amd import ops
    _, wt = _case(1, 3, 3, 96, 160, 5)
    wd = wt.to(dev).contiguous(memory_format=CL)
    wr = ops.conv3x3_x3_rotated(wd)
    assert wr.shape == (96, 160, 3, 3) and wr.is_contiguous(memory_format=CL)
    assert torch.equal(wr.cpu(), wt.flip(2, 3).transpose(0, 1).contiguous())


@pytest.mark.parametrize('n,hw,ci,co', [(4, 14, 64, 128), (2, 28, 128, 128), (8, 7, 256, 256)])
def test_x3conv_through_the_trunk_convolution(dev, n, hw, ci, co, monkeypatch):
    """TrunkConv (the module the client ResNets are built from) with the switch on: forward and data gradient from conv3x3_x3.hip,
    weight gradient from wgrad3x3_x3.hip; all three against fp64 autograd, and against the same module on the library alone."""
    from creamfl_amd import ops
    from creamfl_amd.networks.backbones import TrunkConv
    x, wt = _case(n, hw, hw, ci, co, 11)
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(n, co, hw, hw, generator=g)
    conv = TrunkConv(ci, co, 3, 1, 1).to(dev).to(memory_format=CL)
    with torch.no_grad():
        conv.weight.copy_(wt.to(dev))
    out = {}
    for on in (True, False):
        monkeypatch.setattr(ops, 'X3CONV', [on])
        xd = x.to(dev).contiguous(memory_format=CL).requires_grad_(True)
        conv.weight.grad = None
        before, wbefore = ops.X3CONV_TAKEN[0], ops.X3WGRAD_TAKEN[0]
        y = conv(xd)
        y.backward(gy.to(dev).contiguous(memory_format=CL))
        from creamfl_amd import streams
        streams.join_into_current(dev)
        torch.cuda.synchronize()
        assert ops.X3CONV_TAKEN[0] - before == (2 if on else 0)                  # forward + data gradient
        assert ops.X3WGRAD_TAKEN[0] - wbefore == (1 if on else 0)
        out[on] = (y.detach().cpu().double(), xd.grad.cpu().double(), conv.weight.grad.cpu().double())
        with torch.no_grad():                                                   # the no-grad path (the clients' old model) too
            before = ops.X3CONV_TAKEN[0]
            y2 = conv(xd.detach())
            assert ops.X3CONV_TAKEN[0] - before == (1 if on else 0)
            assert float((y2 - y.detach()).abs().max()) <= 1e-6 * float(y.detach().abs().max()) if on else True
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(gy.double())
    for k, ref in enumerate((y64.detach(), x64.grad, w64.grad)):
        sc = float(ref.abs().max())
        err_x3 = float((out[True][k] - ref).abs().max()) / sc
        err_lib = float((out[False][k] - ref).abs().max()) / sc
        assert err_x3 <= 1e-5, (k, err_x3, err_lib)
        assert err_lib <= 5e-3, (k, err_lib)
