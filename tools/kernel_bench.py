#!/usr/bin/env python3
"""Kernel-level microbenchmarks of the hot-path ops at the SURVEY section 8(d) shapes.
Prints one JSON line per case with per-kernel HIP-event times (library profiler) and the
algorithmic FLOP/byte rates.  GPU only.   python tools/kernel_bench.py [--cases a1,a3,a5,a2,a6]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from creamfl_amd import _lib, ops  # noqa: E402


def unit(*shape, gen=None, device='cuda'):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen, device=device), dim=-1)


def timed(fn, iters=20, warm=3):
    """(wall us per call with NO profiler attached, {kernel: us per launch} from a second, profiled pass).  The profiler
    brackets every launch with two HIP events (~5 us floor per kernel, and it slows the host), so wall time and kernel
    times come from separate passes."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / iters * 1e3
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = {k: round(ms / n * 1e3, 2) for k, (n, ms) in _lib.prof_query().items()}   # us per launch
    return wall, prof


def case_a1(N, D):
    g = torch.Generator(device='cuda').manual_seed(0)
    I = unit(N, D, gen=g).requires_grad_(True)
    T = torch.nn.functional.normalize(I.detach() + 0.5 * unit(N, D, gen=g), dim=-1).requires_grad_(True)
    a = torch.tensor([15.0], device='cuda', requires_grad=True)
    b = torch.tensor([15.0], device='cuda', requires_grad=True)

    def step():
        loss, _ = ops.pair_loss(I, T, a, b)
        loss.backward()
    us, prof = timed(step)
    flops = 3 * 2 * N * N * D
    return {'case': f'a1_pair_loss N={N} D={D}', 'us_per_step': round(us, 1), 'kernels_us': prof,
            'algo_TFLOPs': round(flops / us / 1e6, 2), 'pairs_per_s': round(N / us * 1e6)}


def case_a3(B, M, D, root=True):
    """root: the trainers' form (the loss is the root of the backward pass: pass + finish, no backward launch)."""
    g = torch.Generator(device='cuda').manual_seed(1)
    G = unit(M, D, gen=g)
    Gs = unit(M, D, gen=g)
    idx = torch.randperm(M, device='cuda')[:B]
    f = unit(B, D, gen=g).requires_grad_(True)
    fo = unit(B, D, gen=g)
    from creamfl_amd.algorithms.contrast import client_contrast_loss

    def step():
        loss, _, _ = client_contrast_loss(f, Gs, G, idx, fo, root=root)
        loss.backward()
    us, prof = timed(step)
    flops = 4 * B * M * D
    byts = 2 * M * D * 4
    return {'case': f'a3a4_client_contrast B={B} M={M} D={D}' + ('' if root else ' two-gradient form'), 'us_per_step': round(us, 1),
            'kernels_us': prof, 'kernels_sum_us': round(sum(v for v in prof.values() if v), 1),
            'hbm_frac_of_8TBps': round(M * D * 4 / (sum(v for v in prof.values() if v) * 1e-6) / 8e12, 3),
            'algo_TFLOPs': round(flops / us / 1e6, 2), 'algo_GBps': round(byts / us / 1e3, 1),
            'pairs_per_s': round(B / us * 1e6)}


def case_pool(N=256, H=112, C=64):
    """ResNet stem max pooling (3x3 / 2) on the bench's [N, 64, 112, 112] channels_last bf16 activation, fwd + bwd."""
    x = torch.randn(N, C, H, H, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(N, C, H // 2, H // 2, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def step():
        y = ops.maxpool3s2(x)
        y.backward(gy)
        x.grad = None
    us, prof = timed(step)
    byts = N * H * H * C * 2 + N * (H // 2) ** 2 * C * 3          # each way: big tensor + small tensor + 1-byte taps
    return {'case': f'maxpool3s2 N={N} H={H} C={C}', 'us_per_step': round(us, 1), 'kernels_us': prof,
            'algorithmic_MB_each_way': round(byts / 1e6, 1)}


def case_a5(M, D, C=1, noimg=False):
    """noimg: the 128 x 128 tile GEMM of bank.hip instead of the bank pass (the A/B of the wide kernel at 256 < D <= 512)"""
    g = torch.Generator(device='cuda').manual_seed(2)
    G = unit(M, D, gen=g)
    vecs = [torch.nn.functional.normalize(G + 0.5 * unit(M, D, gen=g), dim=-1) for _ in range(C)]

    def step():
        lp = torch.stack([ops.conw_logprob(v, G) for v in vecs], 0)
        ops.conw_combine(vecs, lp)
    old = ops._CONW_NOIMG
    ops._CONW_NOIMG = bool(noimg) or old
    try:
        us, prof = timed(step, iters=3, warm=1)
    finally:
        ops._CONW_NOIMG = old
    flops = 2 * M * M * D * C
    return {'case': f'a5_conw M={M} D={D} C={C}' + (' tile GEMM' if noimg else ''), 'us_per_step': round(us, 1), 'kernels_us': prof,
            'algo_TFLOPs': round(flops / us / 1e6, 2), 'of_3xbf16_roof': round(flops / us / 1e6 / 833.0, 3)}


def case_a2(N, P, Cd, dh, D, dtype=torch.float32):
    g = torch.Generator(device='cuda').manual_seed(3)
    X = torch.randn(N, P, Cd, generator=g, device='cuda').to(dtype).requires_grad_(True)
    H = torch.randn(N, P, dh, generator=g, device='cuda').to(dtype).requires_grad_(True)
    w2 = (torch.randn(dh, generator=g, device='cuda') / dh ** 0.5).requires_grad_(True)
    out = torch.randn(N, D, generator=g, device='cuda', requires_grad=True)
    rp = torch.randn(N, D, generator=g, device='cuda', requires_grad=True)
    lw = torch.ones(D, device='cuda', requires_grad=True)
    lb = torch.zeros(D, device='cuda', requires_grad=True)

    def step():
        pooled, attn, xm = ops.pie_pool(X, H, w2, None, want_mean=True)
        y, o, r = ops.pie_epilogue(out, rp, lw, lb)
        (pooled.sum() + xm.sum() + y.sum()).backward()
    us, prof = timed(step)
    fwd_bytes = N * P * (Cd + dh) * X.element_size()
    return {'case': f'a2_pie_head N={N} P={P} Cd={Cd} dh={dh} D={D} {str(dtype)[6:]}', 'us_per_step': round(us, 1),
            'kernels_us': prof, 'fwd_algo_MB': round(fwd_bytes / 1e6, 1)}


def case_a6(Nq, Ng, D):
    g = torch.Generator(device='cuda').manual_seed(4)
    img = unit(Nq, D, gen=g)
    cap = torch.nn.functional.normalize(img.repeat_interleave(Ng // Nq, 0) + 1.5 * unit(Ng, D, gen=g), dim=-1)
    ql = torch.arange(Nq, device='cuda')
    gl = torch.arange(Ng, device='cuda') // (Ng // Nq)
    us, prof = timed(lambda: ops.rank_count(img, cap, ql, gl), iters=5, warm=1)
    flops = 2 * Nq * Ng * D                        # the count pass is the one full product (the positives pass skips tiles)
    cnt_us = prof.get('cfl_rank_count_kernel', us)
    return {'case': f'a6_rank Nq={Nq} Ng={Ng} D={D}', 'us_per_step': round(us, 1), 'kernels_us': prof,
            'fp64_TFLOPs_count_kernel': round(flops / cnt_us / 1e6, 2), 'fp64_peak_TFLOPs': 78.6}


def case_f4(B, C, Dw, k=5):
    """Client supervised glue: fused vs the reference's torch op sequence (wall time per fwd+bwd)."""
    g = torch.Generator(device='cuda').manual_seed(4)
    fv = (torch.randn(B, C, generator=g, device='cuda') * 3).requires_grad_(True)
    W = torch.relu(torch.randn(C, Dw, generator=g, device='cuda') * 0.05).requires_grad_(True)
    y = torch.randint(0, C, (B,), generator=g, device='cuda')
    crit = torch.nn.CrossEntropyLoss()
    center = torch.arange(C, device='cuda')

    def fused():
        loss, _ = ops.supervised_glue(fv, y, W, 4.0, k)
        loss.backward()

    def eager():                                   # ClientTrainer.py:344-357 statement by statement
        one_hot = torch.zeros(B, C).scatter_(1, y.cpu().view(-1, 1), 1)
        f2 = fv - 4.0 * one_hot.to('cuda')
        loss = crit(f2, y) + 0.5 * crit(torch.mm(W, W.t()), center)
        _, pred = f2.data.topk(k, 1, True, True)
        correct = pred.t().eq(y.view(1, -1).expand(k, B))
        correct[:1].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / B)
        correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / B)
        loss.backward()
    us, prof = timed(fused)
    us_eager, _ = timed(eager)
    return {'case': f'f4_supervised_glue B={B} C={C} Dw={Dw}', 'us_per_step': round(us, 1), 'kernels_us': prof,
            'eager_torch_us_per_step': round(us_eager, 1)}


def case_bn(N, H, C, res, relu=True):
    """Fused BN(+add)(+ReLU) fwd+bwd on one ResNet-101 activation shape (NHWC bf16): GB/s of each kernel."""
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(N, C, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    r = None
    if res:
        r = torch.randn(N, C, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        r.requires_grad_(True)
    w = torch.ones(C, device='cuda', requires_grad=True)
    b = torch.zeros(C, device='cuda', requires_grad=True)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    dy = torch.randn(N, C, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def step():
        y = ops.bn_act_train(x, w, b, rm, rv, 0.1, 1e-5, relu=relu, residual=r)
        y.backward(dy)
        x.grad = None
        if r is not None:
            r.grad = None
    us, prof = timed(step, iters=10)
    E = N * H * H * C * 2                                   # bytes of one bf16 activation pass
    # bf16 passes per kernel (the ReLU mask is recomputed from x without a residual and is 1 bit / element with one)
    passes = {'cfl_bn_stats_kernel': 1, 'cfl_bn_apply_kernel': 3 if res else 2, 'cfl_bn_bwd_reduce_kernel': 2 + (1 / 16 if res and relu else 0),
              'cfl_bn_bwd_apply_kernel': (4 + 1 / 16 if res else 3)}
    gbps = {k: round(passes[k] * E / v / 1e3) for k, v in prof.items() if k in passes}
    return {'case': f'bn N={N} HxW={H}x{H} C={C} res={int(res)} relu={int(relu)}', 'MB_per_pass': round(E / 1e6, 1),
            'kernels_us': prof, 'kernels_GBps': gbps}


def case_gemm16(H, Ci, Co, variants=(0,), N=256, join=False):
    """1x1 convolution as bf16 NT GEMM on the NHWC-flattened activation vs MIOpen's conv (forward)."""
    import torch.nn.functional as F
    lib = _lib.load()
    M = N * H * H
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(M, Ci, generator=g, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Co, Ci, generator=g, device='cuda') * 0.05).to(torch.bfloat16)
    y = torch.empty(M, Co, device='cuda', dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    ref = (x[:4096].float() @ w.float().t())
    out = {'case': f'gemm16 1x1 {H}x{H} {Ci}->{Co} M={M}', 'roof_us': round(M * (Ci + Co) * 2 / 6.0e6, 1)}
    for var in variants:
        def run():
            _lib.check(lib.cfl_gemm_bf16_nt(x.data_ptr(), Ci, w.data_ptr(), Ci, y.data_ptr(), Co, M, Co, Ci, var, st), 'gemm16')
        try:
            run()
        except Exception:
            continue
        err = (y[:4096].float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        us, prof = timed(run, iters=20)
        k_us = prof.get('cfl_gemm_bf16_kernel', us)
        out[f'v{var}_us'] = k_us
        out[f'v{var}_TF'] = round(2 * M * Ci * Co / k_us / 1e6)
        out[f'v{var}_relerr'] = round(err, 5)
    if join:
        add = torch.randn(M, Co, generator=g, device='cuda').to(torch.bfloat16)
        mask = torch.randint(0, 256, (M * Co // 8,), generator=g, device='cuda', dtype=torch.uint8)
        us, prof = timed(lambda: ops.gemm_bf16_nt(x, w, out=y, add=add, mask=mask), iters=20)
        out['join_us'] = prof.get('cfl_gemm_bf16_kernel', us)
        out['join_roof_us'] = round(M * (Ci + 2 * Co + Co / 16) * 2 / 6.0e6, 1)
        return out
    x4 = x.view(N, H, H, Ci).permute(0, 3, 1, 2)
    w4 = w.view(Co, Ci, 1, 1).contiguous(memory_format=torch.channels_last)
    torch.backends.cudnn.benchmark = True
    us_conv, _ = timed(lambda: F.conv2d(x4, w4), iters=20)
    out['miopen_conv_us'] = round(us_conv, 1)
    return out


def case_wgrad3(H, C, N=256, splits=(0,)):
    """3 x 3 / stride 1 weight gradient (csrc/wgrad3x3.hip) vs MIOpen's kernel at one ResNet-101 shape, bf16 channels_last."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(3)
    cl = torch.channels_last
    x = torch.randn(N, C, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randn(N, C, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
    w = torch.zeros(C, C, 3, 3, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=cl)
    flop = 2.0 * N * H * H * 9 * C * C
    algo = 2.0 * N * H * H * 2 * C + 2.0 * 9 * C * C
    out = {'case': f'wgrad3 {H}x{H} {C}->{C} N={N}', 'gflop': round(flop / 1e9, 1), 'algorithmic_mb': round(algo / 1e6, 1),
           'mfma_floor_us': round(flop / 2.5e9, 1), 'hbm_floor_us': round(algo / 6.3e6, 1)}
    args = (dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    torch.backends.cudnn.benchmark = True
    ref = torch.ops.aten.convolution_backward(*args, [False, True, False])[1]
    us_lib, _ = timed(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]), iters=20)
    out['library_us_wall'] = round(us_lib, 1)
    for sp in splits:
        lib.cfl_conv3x3_wgrad_splits(sp)
        dw = ops.conv3x3_wgrad(dy, x, w)
        if dw is None:
            out['own'] = 'shape not taken'
            break
        err = float((dw.float() - ref.float()).abs().max()) / max(float(ref.float().abs().max()), 1e-9)
        us, prof = timed(lambda: ops.conv3x3_wgrad(dy, x, w), iters=20)
        k = prof.get('cfl_conv3x3_wgrad_kernel', us)
        tag = f's{sp}' if sp else 'default'
        out[f'{tag}_kernel_us'] = k
        out[f'{tag}_reduce_us'] = prof.get('cfl_conv3x3_wgrad_reduce_kernel')
        out[f'{tag}_wall_us'] = round(us, 1)
        out[f'{tag}_TFLOPs'] = round(flop / k / 1e6)
        out[f'{tag}_frac_mfma_peak'] = round(flop / k / 1e6 / 2500.0, 3)
        out[f'{tag}_relerr_vs_library'] = round(err, 5)
    lib.cfl_conv3x3_wgrad_splits(0)
    return out


def case_wgrad1(H, Ci, Co, N=256, wgs=(128,)):
    """1 x 1 weight gradient (csrc/wgrad1x1.hip) vs the library's kernel at one ResNet-101 shape, bf16 channels_last."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(4)
    cl = torch.channels_last
    x = torch.randn(N, Ci, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randn(N, Co, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
    w = torch.zeros(Co, Ci, 1, 1, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=cl)
    M = N * H * H
    flop = 2.0 * M * Ci * Co
    algo = 2.0 * M * (Ci + Co) + 2.0 * Ci * Co
    out = {'case': f'wgrad1 {H}x{H} {Ci}->{Co} N={N}', 'gflop': round(flop / 1e9, 1), 'algorithmic_mb': round(algo / 1e6, 1),
           'hbm_floor_us': round(algo / 6.3e6, 1)}
    args = (dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)
    torch.backends.cudnn.benchmark = True
    ref = torch.ops.aten.convolution_backward(*args, [False, True, False])[1]
    us_lib, _ = timed(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]), iters=20)
    out['library_us_wall'] = round(us_lib, 1)
    old = lib.cfl_conv1x1_wgrad_workgroups(0)
    for n in wgs:
        lib.cfl_conv1x1_wgrad_workgroups(n)
        dw = ops.conv1x1_wgrad(dy, x, w)
        if dw is None:
            out['own'] = 'shape not taken'
            break
        err = float((dw.float() - ref.float()).abs().max()) / max(float(ref.float().abs().max()), 1e-9)
        us, prof = timed(lambda: ops.conv1x1_wgrad(dy, x, w), iters=20)
        k = prof.get('cfl_conv1x1_wgrad_kernel', us)
        out[f'wg{n}_kernel_us'] = k
        out[f'wg{n}_reduce_us'] = prof.get('cfl_conv1x1_wgrad_reduce_kernel')
        out[f'wg{n}_wall_us'] = round(us, 1)
        out[f'wg{n}_algorithmic_TBps'] = round(algo / k / 1e6, 2)
        out[f'wg{n}_relerr_vs_library'] = round(err, 5)
    lib.cfl_conv1x1_wgrad_workgroups(old)
    return out


def case_x3conv(H, C, N=128, dbg=False):
    """3 x 3 / stride 1 convolution of fp32 channels_last tensors: csrc/conv3x3_x3.hip (3 x bf16 split on the bf16 matrix pipe) vs the
    library's fp32 kernels, forward and data gradient, at one BasicBlock shape of the clients' ResNet-18 (batch 128); errors of both
    against fp64 on two images."""
    import torch.nn.functional as F
    cl = torch.channels_last
    g = torch.Generator(device='cuda').manual_seed(H + C)
    x = torch.randn(N, C, H, H, generator=g, device='cuda').contiguous(memory_format=cl)
    w = (torch.randn(C, C, 3, 3, generator=g, device='cuda') / (3.0 * C ** 0.5)).contiguous(memory_format=cl)
    dy = torch.randn(N, C, H, H, generator=g, device='cuda').contiguous(memory_format=cl)
    flop = 2.0 * N * H * H * C * C * 9
    out = {'case': f'x3conv {H}x{H} {C}->{C} N={N} fp32 channels_last', 'gflop': round(flop / 1e9, 1), 'gflop_on_the_bf16_pipe': round(3 * flop / 1e9, 1),
           'mfma_floor_us': round(3 * flop / 2.5e15 * 1e6, 1)}
    with torch.backends.cudnn.flags(enabled=True, benchmark=True):
        for _ in range(3):
            F.conv2d(x, w, None, 1, 1)
            torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
        us, _ = timed(lambda: F.conv2d(x, w, None, 1, 1), iters=20)
        out['library_fwd_us'] = round(us, 1)
        us, _ = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]), iters=20)
        out['library_dgrad_us'] = round(us, 1)
        us, _ = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]), iters=20)
        out['library_wgrad_us'] = round(us, 1)
        ylib = F.conv2d(x[:2], w, None, 1, 1)
    ref = F.conv2d(x[:2].double(), w.double(), None, 1, 1)
    sc = float(ref.abs().max())
    out['library_fwd_relerr_vs_fp64'] = float((ylib.double() - ref).abs().max()) / sc
    M = N * H * H
    VS = (522, 521, 822, 821, 842, 522, 521, 822, 821, 0) + ((222, 1222, 2222, 3222, 4222, 6222) if dbg else ())
    for v in VS:
        if v % 10 == 2 and C % 128:
            continue
        y = ops.conv3x3_x3_forward(x, w, v)
        if v < 1000:
            out[f'x3_v{v}_relerr_vs_fp64'] = float((y[:2].double() - ref).abs().max()) / sc
        us, prof = timed(lambda: ops.conv3x3_x3_forward(x, w, v), iters=20)
        k = prof.get('cfl_conv3x3_x3_kernel', us)                   # (version 3: without the weight-image launch, ~3 us)
        out.setdefault(f'x3_v{v}_fwd_us', []).append(k)                  # (a variant listed twice is timed twice: A/B/A/B on one lease)
        out[f'x3_v{v}_TFLOPs_fp32_equivalent'] = round(flop / k / 1e6)
    wg = w.clone().requires_grad_(True)                                      # (a trainable weight: its image is rebuilt per call)
    us, prof = timed(lambda: ops.conv3x3_x3_forward(dy, wg, rotated=True), iters=20)
    out['x3_dgrad_us_incl_weight_rotation'] = round(sum(v for v in prof.values() if v), 1)
    dwl = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw = ops.conv3x3_x3_wgrad(dy, x, w)
    if dw is not None:
        # errors against fp64 (CPU) on the first four images
        xs, dys = x[:4].contiguous(memory_format=cl), dy[:4].contiguous(memory_format=cl)
        ref64 = torch.ops.aten.convolution_backward(dys.double().cpu(), xs.double().cpu(), w.double().cpu(), None, [1, 1], [1, 1], [1, 1], False,
                                                    [0, 0], 1, [False, True, False])[1]
        sc64 = float(ref64.abs().max())
        out['x3_wgrad_relerr_vs_fp64'] = float((ops.conv3x3_x3_wgrad(dys, xs, w).double().cpu() - ref64).abs().max()) / sc64
        dwl4 = torch.ops.aten.convolution_backward(dys, xs, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        out['library_wgrad_relerr_vs_fp64'] = float((dwl4.double().cpu() - ref64).abs().max()) / sc64
        out['x3_vs_library_wgrad_full_batch'] = float((dw - dwl).abs().max() / dwl.abs().max())
        us, prof = timed(lambda: ops.conv3x3_x3_wgrad(dy, x, w), iters=20)
        out['x3_wgrad_kernel_us'] = prof.get('cfl_conv3x3_x3_wgrad_kernel')
        out['x3_wgrad_reduce_us'] = prof.get('cfl_conv3x3_x3_wgrad_reduce_kernel')
        out['x3_wgrad_us'] = round(sum(v for v in prof.values() if v), 1)
        out['speedup_wgrad'] = round(out['library_wgrad_us'] / out['x3_wgrad_us'], 2)
        lib = ops._lib.load()
        for sp in (64, 128, 256, 512):
            old = lib.cfl_conv3x3_x3_wgrad_splits(sp)
            try:
                us, prof = timed(lambda: ops.conv3x3_x3_wgrad(dy, x, w), iters=10)
            finally:
                lib.cfl_conv3x3_x3_wgrad_splits(old)
            out[f'x3_wgrad_splits{sp}_us'] = [prof.get('cfl_conv3x3_x3_wgrad_kernel'), prof.get('cfl_conv3x3_x3_wgrad_reduce_kernel')]
    best = min(min(out[f'x3_v{v}_fwd_us']) for v in VS if v < 1000 and f'x3_v{v}_fwd_us' in out)
    out['best_variant'] = min((min(out[f'x3_v{v}_fwd_us']), v) for v in VS if v < 1000 and f'x3_v{v}_fwd_us' in out)[1]
    out['speedup_fwd'] = round(out['library_fwd_us'] / best, 2)
    out['speedup_dgrad'] = round(out['library_dgrad_us'] / out['x3_dgrad_us_incl_weight_rotation'], 2)
    return out


def case_opt(cnn='resnet101'):
    """fused clip + AdamP over the real parameter set of the bench model (ResNet-101 + BERT-base PCME)."""
    from creamfl_amd.algorithms.optimizers import AdamP
    from creamfl_amd.networks.models import get_model
    from creamfl_amd.utils.config import default_config
    torch.manual_seed(0)
    model = get_model({'<pad>': 0}, default_config(embed_dim=512, cnn_type=cnn).model, False).cuda()
    model.to(memory_format=torch.channels_last)
    params = [p for p in model.parameters()]
    for p in params:
        p.grad = torch.randn_like(p) * 1e-3
    opt = AdamP(params, lr=2e-4)
    n = sum(p.numel() for p in params)
    us, prof = timed(lambda: opt.step(clip=(params, 2.0)), iters=10, warm=2)
    return {'case': f'opt_clip_adamp {cnn}+bert-base n_params={n}', 'us_per_step': round(us, 1), 'kernels_us': prof,
            'pass1_GBps': round(24 * n / prof['cfl_adamp_pass1_kernel'] / 1e3, 1),
            'pass3_GBps': round(16 * n / prof['cfl_adamp_pass3_kernel'] / 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='a1,a3,a5,a2,a6,f4')
    ap.add_argument('--conw-m', type=int, default=50000)
    args = ap.parse_args()
    cases = args.cases.split(',')
    out = []
    if 'a1' in cases:
        out += [case_a1(256, 512), case_a1(128, 256), case_a1(4096, 512)]
    if 'a1big' in cases:
        out += [case_a1(4096, 512)]
    if 'a3' in cases:
        out += [case_a3(128, 50000, 256), case_a3(128, 50000, 256, root=False), case_a3(256, 50000, 512), case_a3(128, 50000, 768),
                case_a3(32, 50000, 256)]
    if 'pool' in cases:
        out += [case_pool()]
    if 'a3one' in cases:
        out += [case_a3(128, 50000, 256)]
    if 'a5' in cases:
        out += [case_a5(args.conw_m, 256)]
    if 'a5wide' in cases:            # D = 512 (configs[1]): the 4-wave wide bank kernel vs the 128 x 128 tile GEMM of bank.hip; 768 (configs[4]): tile GEMM
        out += [case_a5(args.conw_m, 512), case_a5(args.conw_m, 512, noimg=True), case_a5(args.conw_m, 384), case_a5(args.conw_m, 768),
                case_a5(args.conw_m, 768, noimg=True)]
    if 'a2' in cases:
        out += [case_a2(256, 49, 2048, 1024, 512), case_a2(256, 49, 2048, 1024, 512, torch.bfloat16), case_a2(128, 49, 512, 256, 256)]
    if 'a6' in cases:
        out += [case_a6(1000, 5000, 512), case_a6(5000, 25000, 512)]
    if 'f4' in cases:
        out += [case_f4(512, 100, 512), case_f4(512, 10, 512), case_f4(512, 4, 512, 4)]
    if 'bn' in cases:
        for (H, C, res) in [(112, 64, False), (56, 64, False), (56, 256, True), (56, 128, False), (28, 128, False),
                            (28, 512, True), (28, 256, False), (14, 256, False), (14, 1024, True), (14, 512, False),
                            (7, 512, False), (7, 2048, True)]:
            out.append(case_bn(256, H, C, res))
    if 'gemm16' in cases:
        os.environ.setdefault('MIOPEN_FIND_MODE', '2')
        for (H, Ci, Co) in [(56, 64, 64), (56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128),
                            (14, 256, 1024), (14, 1024, 256), (7, 512, 2048), (7, 2048, 512)]:
            vs = [v for v in (90, 44, 24, 42, 22, 21, 41) if not (v in (44, 42, 22) and Co < 128) and not (v == 24 and Co < 256)]
            out.append(case_gemm16(H, Ci, Co, vs))
    if 'fwdstats16' in cases:
        # forward 1x1 convolution + the statistics pass of the BatchNorm behind it: library convolution + cfl_bn_stats vs the
        # B-resident GEMM with the statistics in its epilogue (cfl_gemm_bf16_nt_stats)
        import torch.nn.functional as F
        lib = _lib.load()
        torch.backends.cudnn.benchmark = True
        for (H, Ci, Co) in [(56, 64, 256), (28, 128, 512), (14, 256, 1024)]:
            M = 256 * H * H
            g = torch.Generator(device='cuda').manual_seed(5)
            x = torch.randn(256, Ci, H, H, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(Co, Ci, 1, 1, generator=g, device='cuda') * 0.05).to(torch.bfloat16)
            y = torch.empty(256, Co, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
            nblk = lib.cfl_gemm_bf16_nt_stats_nblk(M, Co, Ci)
            ps = torch.empty(2 * nblk * Co, device='cuda', dtype=torch.float32)
            st = torch.cuda.current_stream().cuda_stream
            us_f, prof = timed(lambda: _lib.check(lib.cfl_gemm_bf16_nt_stats(x.data_ptr(), Ci, w.data_ptr(), Ci, y.data_ptr(), M, Co, Ci,
                                                                             ps.data_ptr(), st), 'stats'), iters=20)
            bnw = torch.ones(Co, device='cuda'); bnb = torch.zeros(Co, device='cuda')
            rm = torch.zeros(Co, device='cuda'); rv = torch.ones(Co, device='cuda')

            def lib_path():
                z = F.conv2d(x, w)
                return ops.bn_act_train(z, bnw, bnb, rm, rv, 0.1, 1e-5, relu=False)
            us_l, prof_l = timed(lib_path, iters=20)
            out.append({'case': f'fwdstats16 {H}x{H} {Ci}->{Co}', 'gemm_stats_us': prof.get('cfl_gemm_bf16_kernel', us_f),
                        'lib_conv_plus_bn_us_wall': round(us_l, 1), 'bn_stats_us': prof_l.get('cfl_bn_stats_kernel'),
                        'bn_apply_us': prof_l.get('cfl_bn_apply_kernel')})
    if 'dgrad16' in cases:
        # the data gradients of the trunk's 1x1 convolutions as the step runs them: K = forward Co, N = forward Ci; conv1 of a
        # block with the gradient join in the epilogue, conv3 plain (B-resident kernel vs the tile kernel: CFL_GEMM_NO_BRES=1)
        for (H, Ci, Co) in [(56, 64, 256), (28, 128, 512), (14, 256, 1024), (7, 512, 2048)]:
            out.append(case_gemm16(H, Ci, Co, (0,), join=True))
        for (H, Ci, Co) in [(56, 256, 64), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)]:
            out.append(case_gemm16(H, Ci, Co, (0,)))
    if 'wgrad3' in cases:
        out.append(case_wgrad3(14, 256, splits=(0, 16)))
        out.append(case_wgrad3(7, 512, splits=(0,)))
        out.append(case_wgrad3(28, 128, splits=(0, 64)))
        out.append(case_wgrad3(56, 64, splits=(0, 128)))
    if 'w3dbg' in cases:
        # switch-off decomposition of the 3 x 3 weight-gradient kernel at layer3's shape (results of modes 1 / 2 are wrong on purpose)
        lib = _lib.load()
        g = torch.Generator(device='cuda').manual_seed(3)
        cl = torch.channels_last
        x = torch.randn(256, 256, 14, 14, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
        dy = torch.randn(256, 256, 14, 14, generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=cl)
        w = torch.zeros(256, 256, 3, 3, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=cl)
        rec = {'case': 'w3dbg 14x14 256->256 N=256'}
        for mode, name in ((0, 'full'), (1, 'no_staging'), (2, 'staging_and_barriers_only')):
            lib.cfl_conv3x3_wgrad_debug(mode)
            us, prof = timed(lambda: ops.conv3x3_wgrad(dy, x, w), iters=20)
            rec[name + '_us'] = prof.get('cfl_conv3x3_wgrad_kernel', us)
        lib.cfl_conv3x3_wgrad_debug(0)
        out.append(rec)
    if 'x3conv' in cases:
        for H, C in ((28, 128), (56, 64), (14, 256), (7, 512)):
            out.append(case_x3conv(H, C, dbg='x3dbg' in cases))
    if 'wgrad1' in cases:
        for (H, Ci, Co) in [(14, 1024, 256), (14, 256, 1024), (28, 512, 128), (28, 128, 512), (56, 256, 64), (56, 64, 256), (56, 64, 64),
                            (7, 2048, 512), (7, 512, 2048)]:
            out.append(case_wgrad1(H, Ci, Co, wgs=(128, 256, 64)))
    if 'opt' in cases:
        out += [case_opt()]
    for r in out:
        print(json.dumps(r))


if __name__ == '__main__':
    main()
