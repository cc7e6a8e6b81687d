"""HIP-graph capture of launch-bound training steps.

The client contrast step (src/algorithms/ClientTrainer.py:376-421: features -> inter / intra contrast against the frozen global
banks -> backward -> SGD step) is a few hundred small launches per batch on a ResNet-18-sized model; the contrast terms
themselves are 3 kernels (44 us at B = 128, D = 256) behind ~135 us of Python / autograd / ctypes work per step.  The C ABI
allocates nothing and never synchronises, so the WHOLE step replays from one hipGraph: `GraphedStep` runs the first calls
eagerly (they are real training steps: the libraries pick their kernels, optimizers create their state, the bank images are
built), captures the next call, and from then on copies each batch into the captured input tensors and replays.

Contract of the wrapped function: fixed tensor shapes / dtypes, every input a device tensor (or copied into one here), no host
synchronisation and no data-dependent Python control flow inside, outputs = tensors (returned as the graph's static outputs:
valid until the next call).  A call whose input shapes differ from the captured ones (the ragged last batch of an epoch) runs
eagerly.

One more condition, found the hard way (docs/history/tools/mm_graph_probe.py): when the capture happens, NO loss of an earlier eager step on the
legacy default stream may still be alive.  A live loss keeps its autograd graph's AccumulateGrad nodes alive; a node belongs to the
stream it was made on; the captured backward makes that stream wait for the capturing one -- and the HIP runtime does not turn the
default stream's event record into a capture node, so the fork is never joined and hipStreamEndCapture faults (a segmentation
fault, not an error code: the fallback below cannot catch it).  Trainers keep such steps in functions of their own
(MMClientTrainer._local_epoch) so that their locals are dead; nodes made on any other stream are joined properly.
The condition is ENFORCED, not only documented: before a capture `live_grad_nodes` probes the AccumulateGrad node of every parameter
the step updates (`guard_params`, by default the optimizer's); if one is alive -- after a `gc.collect()` -- the step stays eager and
says so.  (A node's stream cannot be read from Python, so the check is conservative: any live node refuses the capture.  The wrapped
functions return detached losses, so nothing of a well-formed caller is alive here.)
"""
import gc
import os
import warnings

import torch


def live_grad_nodes(params):
    """How many of `params` have an AccumulateGrad node that some autograd graph (a loss that is still referenced, with or
    without its buffers) keeps alive.  A tensor holds its accumulator weakly: asking for the gradient edge creates a node that dies
    with the handle unless a graph owns it.  So: tag the node's metadata, drop the handle, ask again -- the tag survives only on a
    node that something else holds."""
    tok = object()
    n_live = 0
    for p in params:
        if not (torch.is_tensor(p) and p.requires_grad and p.is_leaf):
            continue
        node = torch.autograd.graph.get_gradient_edge(p).node
        node.metadata['_cfl_probe'] = tok
        del node
        node = torch.autograd.graph.get_gradient_edge(p).node
        if node.metadata.get('_cfl_probe') is tok:
            n_live += 1
            del node.metadata['_cfl_probe']
        del node
    return n_live


class GraphedStep:
    def __init__(self, fn, warmup=3, enabled=True, log=None, optimizer=None, other_threads=False, guard_params=None):
        """guard_params: the parameters the step updates (default: those of `optimizer`, any torch optimizer): no AccumulateGrad node
        of theirs may be alive when the capture starts (module docstring) -- checked, the step stays eager otherwise.
        log: callable(str) that is told ONCE why a capture failed (default: warnings.warn) -- a step that silently stays eager
        looks like a performance regression with no trace.
        optimizer: an optimizer whose step needs to know about captures and replays (creamfl_amd's fused AdamP: the step count of
        its bias corrections lives on the device inside a graph, `prepare_capture` / `capture_begin` / `capture_end`, and the
        host's counts follow through the handle's `replayed()`; the handle's `valid()` is asked before every replay)."""
        self.fn = fn
        self.log = log
        # other_threads: another thread of the process issues HIP work while this step is captured (utils.prefetch's copy thread
        # stages or generates the next batches): capture in thread-local mode -- the default global mode fails ANY thread's
        # allocation for the duration of the capture
        self.capture_mode = os.environ.get('CFL_GRAPH_CAPTURE_MODE') or ('thread_local' if other_threads else 'global')
        if guard_params is None and hasattr(optimizer, 'param_groups'):
            guard_params = [p for g in optimizer.param_groups for p in g['params']]
        self.guard_params = list(guard_params) if guard_params is not None else []
        self.optimizer = optimizer if hasattr(optimizer, 'capture_begin') else None
        self._opt_handle = None
        if self.optimizer is not None and bool(enabled) and torch.cuda.is_available():
            self.optimizer.prepare_capture()
        self._side = None
        self.warmup = max(1, int(warmup))
        self.enabled = bool(enabled) and torch.cuda.is_available()
        self.calls = 0
        self.replays = 0
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.sig = None
        self.failed = None

    @staticmethod
    def _signature(inputs):
        return tuple((tuple(t.shape), t.dtype) for t in inputs)

    def _copy_in(self, inputs):
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)

    def _capture(self, inputs):
        if self.guard_params:
            n_live = live_grad_nodes(self.guard_params)
            if n_live:
                gc.collect()                                      # (a loss that only a reference cycle holds)
                n_live = live_grad_nodes(self.guard_params)
            if n_live:
                # hipStreamEndCapture would FAULT (not raise) if such a node was made on the legacy default stream: never try
                self._stay_eager('%d parameter(s) of the step still have a live AccumulateGrad node -- a loss of an earlier eager step is '
                                 'referenced somewhere (return detached losses, keep eager steps in functions of their own)' % n_live)
                return False
        self.static_in = [torch.empty_like(t) if t.is_cuda else torch.empty(t.shape, dtype=t.dtype, device=self._device) for t in inputs]
        self._copy_in(inputs)
        self.sig = self._signature(inputs)
        graph = torch.cuda.CUDAGraph()
        handle = self.optimizer.capture_begin() if self.optimizer is not None else None
        try:
            with torch.cuda.graph(graph, capture_error_mode=self.capture_mode):
                out = self.fn(*self.static_in)
        except Exception as e:                                    # noqa: BLE001  (capture not possible: stay eager, say why once)
            if handle is not None:
                self.optimizer.capture_end(handle, ok=False)     # the recorded step never ran: the host's step counts go back
            self._stay_eager(repr(e)[:300])
            return False
        if handle is not None:
            self.optimizer.capture_end(handle)
        self.graph, self.static_out, self._opt_handle = graph, out, handle
        return True

    def _stay_eager(self, why):
        self.failed = why
        self.enabled = False
        self.graph = self.static_out = self.static_in = self._opt_handle = None    # (the graph's pool and its inputs return)
        torch.cuda.synchronize()
        msg = 'GraphedStep: no HIP graph (capture failed or went stale), the step stays eager: %s' % self.failed
        (self.log or warnings.warn)(msg)

    def _warm(self, inputs):
        """An eager warm-up step on a SIDE stream (the capture recipe: the libraries bind their handles and workspaces, and the
        allocator its blocks, away from the stream the rest of the program runs on); the caller's stream waits for it."""
        cur = torch.cuda.current_stream(self._device)
        if self._side is None:
            self._side = torch.cuda.Stream(self._device)
        args = [t.to(self._device, non_blocking=True) for t in inputs]
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            out = self.fn(*args)
        cur.wait_stream(self._side)
        for t in args + [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o)]:
            if t.is_cuda:
                t.record_stream(cur)
        return out

    def __call__(self, *inputs, device=None):
        """inputs: tensors (host or device).  Returns fn's outputs."""
        self.calls += 1
        self._device = device if device is not None else next((t.device for t in inputs if t.is_cuda), torch.device('cuda'))
        if not self.enabled:
            return self.fn(*[t.to(self._device, non_blocking=True) for t in inputs])
        if self.graph is None:
            if self.calls <= self.warmup:
                return self._warm(inputs)                          # eager: real steps
            if not self._capture(inputs):
                return self.fn(*[t.to(self._device, non_blocking=True) for t in inputs])
            self.graph.replay()                                   # capture records, it does not execute: this runs the step
            self.replays += 1
            return self.static_out
        if self._signature(inputs) != self.sig:
            return self.fn(*[t.to(self._device, non_blocking=True) for t in inputs])               # ragged batch: eager
        if self._opt_handle is not None:
            if not self._opt_handle.valid():
                # another step updated a different set of parameters since the capture: the captured step-count offsets are void
                self._stay_eager('the optimizer stepped a different parameter set since the capture')
                return self.fn(*[t.to(self._device, non_blocking=True) for t in inputs])
            self._opt_handle.replayed()
        self._copy_in(inputs)
        self.graph.replay()
        self.replays += 1
        return self.static_out
