"""Auxiliary HIP streams of the training step, one set per device.

The step's main stream carries the image tower; independent work runs beside it:
  'text'  the text tower, forward and (autograd replays a backward op on the stream of its forward) backward
          (networks/models/pcme.py)
  'wgrad' the weight gradients of the trunk convolutions, which nothing on the critical path waits for (ops.conv1x1)
Kept at module level (not on nn.Modules, which must stay deepcopy-able).  Every consumer of gradients that may come from
these streams -- the optimizer step, DDP's bucket all-reduce -- first calls `join_into_current`."""
import ctypes
import os

import torch

_STREAMS = {}
_RAW = []                    # CU-masked HIP streams created here (kept for the life of the process)


def _cu_masked_stream(device, spec):
    """A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), wrapped for torch.
    spec = 'n' (n of every 256 CUs, spread evenly over the mask) or '0x...' (the mask itself, bit i = CU i).
    Measurement knob of round 6 (CFL_WGRAD_CUS): do the side stream's weight-gradient kernels cost the main stream less when
    they cannot take every CU?"""
    if spec.lower().startswith('0x'):
        mask = int(spec, 16)
    else:
        n = max(1, min(256, int(spec)))
        mask, acc = 0, 0
        for i in range(256):                  # Bresenham spread: n bits set among 256
            acc += n
            if acc >= 256:
                acc -= 256
                mask |= 1 << i
    words = (ctypes.c_uint32 * 8)(*[(mask >> (32 * i)) & 0xffffffff for i in range(8)])
    hip = ctypes.CDLL('libamdhip64.so')
    raw = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), ctypes.c_uint32(8), words)
    if rc != 0 or not raw.value:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
    _RAW.append(raw)
    return torch.cuda.ExternalStream(raw.value, device=device)


def get(device, name):
    key = (torch.device(device), name)
    s = _STREAMS.get(key)
    if s is None:
        spec = os.environ.get('CFL_%s_CUS' % name.upper())
        s = _STREAMS[key] = _cu_masked_stream(device, spec) if spec else torch.cuda.Stream(device=device)
        # Gradients of the text tower / the weight gradients are PRODUCED on these streams on purpose and joined before
        # anything consumes them (join_into_current): autograd's "AccumulateGrad stream mismatch" warning is expected here.
        warn_off = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if warn_off is not None:
            warn_off(False)
    return s


def existing(device):
    device = torch.device(device)
    return [s for (d, _), s in _STREAMS.items() if d == device]


def join_into_current(device):
    """Make the current stream of `device` wait for all work queued so far on the auxiliary streams."""
    cur = torch.cuda.current_stream(device)
    for s in existing(device):
        if s != cur:
            cur.wait_stream(s)


# ---- deferred weight gradients --------------------------------------------------------------------------------------
# A convolution's weight gradient is off the critical path, but when it runs concurrently with the (LDS/MFMA-bound)
# data-gradient GEMMs it slows exactly the kernels that ARE on the critical path.  The backward of a convolution
# therefore only queues its weight-gradient launch; the queue is flushed onto the 'wgrad' stream when the main stream
# enters a long HBM-bound phase (the backward of a residual BatchNorm) and at the end of the backward pass.
# Multi-GPU: the deferred task reports its finished gradient to the gradient buckets itself (GRAD_READY, dist.py).
DEFER_WGRAD = [True]
GRAD_READY = [None]          # callback(parameter): a deferred weight gradient has been written (dist.GradBuckets.notify)
_PENDING = []
_PENDING_FOR = [-1]


def defer(task):
    gid = torch._C._current_graph_task_id()
    if gid != _PENDING_FOR[0]:
        del _PENDING[:]                  # leftovers of a backward pass that was aborted: their gradients are void
        _PENDING_FOR[0] = gid
    _PENDING.append(task)


FLUSH_POLICY = [0]           # measurement knob (tools/ab_step.py): 0 = everything at a residual BatchNorm backward (default);
#                              1 = everything at EVERY BatchNorm backward; 2 = one queued task per BatchNorm backward;
#                              n >= 3 = at a residual BatchNorm backward once n tasks are queued


def flush(device, limit=None):
    """Launch the queued weight gradients (all, or the `limit` oldest) on the 'wgrad' stream, after everything queued so far
    on the current one."""
    if not _PENDING:
        return
    main = torch.cuda.current_stream(device)
    side = get(device, 'wgrad')
    side.wait_stream(main)
    n = len(_PENDING) if limit is None else min(limit, len(_PENDING))
    tasks = list(_PENDING[:n])
    del _PENDING[:n]
    with torch.cuda.stream(side):
        for t in tasks:
            t(main, side)
