"""Auxiliary HIP streams of the training step, one set per device.

The step's main stream carries the image tower; independent work runs beside it:
  'text'  the text tower, forward and (autograd replays a backward op on the stream of its forward) backward
          (networks/models/pcme.py)
  'wgrad' the weight gradients of the trunk convolutions, which nothing on the critical path waits for (ops.conv1x1)
Kept at module level (not on nn.Modules, which must stay deepcopy-able).  Every consumer of gradients that may come from
these streams -- the optimizer step, DDP's bucket all-reduce -- first calls `join_into_current`."""
import torch

_STREAMS = {}


def get(device, name):
    key = (torch.device(device), name)
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=device)
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
