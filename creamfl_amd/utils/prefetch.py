"""Host -> device staging of loader batches one (or two) batches ahead of the training step.

The reference's loops move every batch synchronously at the top of the iteration (`images = images.to(self.device)`,
src/algorithms/retrieval_trainer.py:192-206, MMFL.py:346-351, ClientTrainer.py:376-380): 154 MB of fp32 images per server
batch, ~3 ms over PCIe plus -- for pageable memory -- a staging copy on the host thread that also has to enqueue the step's
~2000 kernel launches.  `DevicePrefetcher` wraps any loader with the reference's batch-tuple contract
(src/datasets/_dataloader.py:49-64): a daemon thread takes batches from the loader, pins the tensors that are not pinned yet,
issues their copies on a dedicated HIP stream and hands (device batch, event) to the consumer through a bounded queue; the
consumer's stream waits for the event (no host synchronisation anywhere).  Non-tensor items (caption strings, ids, index lists)
pass through; batches that already live on the device pass through untouched.
"""
import os
import queue
import threading

import torch

_INLINE = [os.environ.get('CFL_PREFETCH_INLINE', '0') == '1']      # measurement knob: no producer thread

_H2D = {}        # device -> copy stream.  NOT one of streams.py's auxiliary streams: the end-of-backward join and the gradient
                 # reducer wait for those, and must not wait for the next batch's copy


def _copy_stream(device):
    s = _H2D.get(device)
    if s is None:
        s = _H2D[device] = torch.cuda.Stream(device=device)
    return s


class DevicePrefetcher:
    def __init__(self, loader, device, depth=2, pin=True):
        self.loader = loader
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None and torch.cuda.is_available():
            self.device = torch.device('cuda', torch.cuda.current_device())     # the copy thread must use the caller's device
        self.depth = max(1, int(depth))
        self.pin = pin
        self.dataset = getattr(loader, 'dataset', None)
        self._pool = {}

    def __len__(self):
        return len(self.loader)

    def _pinned_like(self, item, slot):
        """A reusable pinned staging buffer for pageable tensors of this shape (hipHostMalloc per batch costs more than the copy:
        154 MB of images pinned afresh every step held the loop at 0.88 of the resident step).  `slot` cycles over depth + 2
        buffers; a buffer is reused only after the copy that last read it has completed (its event)."""
        key = (slot, tuple(item.shape), item.dtype)
        ent = self._pool.get(key)
        if ent is None:
            ent = self._pool[key] = [torch.empty(item.shape, dtype=item.dtype, pin_memory=True), None]
        if ent[1] is not None:
            ent[1].synchronize()                     # (copy thread only: the training thread never waits here)
        return ent

    def _stage(self, batch, stream, slot=0):
        out, moved, used = [], False, []
        with torch.cuda.stream(stream):
            for pos, item in enumerate(batch):
                host_lens = None
                if pos == 3 and torch.is_tensor(item) and item.dim() == 1 and not item.is_floating_point():
                    # caption_lens of the batch-tuple contract (src/datasets/_dataloader.py:49-64): keep the host's copy with the
                    # device tensor -- the packed text tower plans on it without reading the device (PCME._pack_plan)
                    host_lens = getattr(item, '_cfl_host_lens', None)
                    if host_lens is None and not item.is_cuda:
                        host_lens = tuple(item.tolist())
                if torch.is_tensor(item) and not item.is_cuda:
                    if self.pin and not item.is_pinned() and item.numel() >= 4096:
                        ent = self._pinned_like(item, slot)
                        ent[0].copy_(item)
                        src = ent[0]
                        used.append(ent)
                    else:
                        src = item
                    item = src.to(self.device, non_blocking=True)    # (a loader's own pinned tensors: the host allocator keeps them alive)
                    moved = True
                if host_lens is not None:
                    item._cfl_host_lens = host_lens
                out.append(item)
        ev = None
        if moved:
            ev = torch.cuda.Event()
            ev.record(stream)
            for ent in used:
                ent[1] = ev
        return tuple(out) if isinstance(batch, tuple) else out, ev

    def __iter__(self):
        if self.device.type != 'cuda':
            yield from self.loader
            return
        if _INLINE[0]:
            # measurement form: the loader iterated on the consumer's thread.  Measured SLOWER even for loaders whose batches are born in
            # HBM (tools/federation_step_trace.py, profiles/r6_federation_step_trace.jsonl: 30.2 vs 28.2 ms per public batch): the
            # producer thread's ~10 launches per batch overlap the host-bound step's GIL-free stretches (ctypes calls release the GIL)
            for batch in self.loader:
                staged, ev = self._stage(batch, h2d_stream := _copy_stream(self.device))
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in staged:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                yield staged
            return
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        h2d = _copy_stream(self.device)

        def put(x):
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                if self.device.index is not None:
                    torch.cuda.set_device(self.device)       # (a new thread starts on device 0)
                for n, batch in enumerate(self.loader):
                    if not put(self._stage(batch, h2d, n % (self.depth + 2))):
                        return
                put(None)
            except BaseException as e:          # re-raised in the consumer
                put(e)

        th = threading.Thread(target=worker, name='creamfl-h2d', daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in batch:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)         # allocated on the copy stream, consumed on this one
                yield batch
        finally:
            stop.set()                                   # an early `break` in the consumer must not leave the thread blocked
