"""Mirror of src/algorithms/optimizers.py:7-58 (get_optimizer / get_lr_scheduler).

`AdamP` implements the published algorithm of adamp==0.3.0 (Heo et al., "AdamP: Slowing Down the
Slowdown for Momentum Optimizers on Scale-invariant Weights", ICLR 2021), which the reference imports as
a third-party package that is not vendored and not installed here: PARITY UNPINNED -- the HIP kernels are
checked against the paper restatement in oracle/adamp.py (tests/test_gpu_optimizer.py), not against the
package.  The projection test `cosine_sim.max() < delta / sqrt(dim)` is evaluated on the device, so a step
issues no host synchronisation (the package syncs once or twice per parameter tensor).
"""
import ctypes
import warnings
import weakref

import numpy as np
import torch
import torch.optim as optim
from torch.optim.optimizer import Optimizer

from .. import _lib


class AdamP(Optimizer):
    """AdamP on the HIP path: gradient clipping + the whole optimizer step as three multi-tensor kernels
    (csrc/adamp.hip), no host synchronisation.  Same constructor and state_dict layout as adamp.AdamP
    (`step`, `exp_avg`, `exp_avg_sq` per parameter).  Parameters must live on a HIP device when `step()` runs;
    there is no CPU path (the paper restatement used as the test oracle is oracle/adamp.py)."""

    ROWS_TARGET = 16384          # elements per work item

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, delta=0.1, wd_ratio=0.1,
                 nesterov=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, delta=delta, wd_ratio=wd_ratio,
                        nesterov=nesterov)
        super().__init__(params, defaults)
        self._plans = {}
        self.grad_override = None        # {parameter: tensor to read its gradient from} (multi-GPU: dist.GradBuckets views)
        self.grad_override_consume = None   # callable: raises unless those views hold THIS backward pass's averages
        # steps replayed from a HIP graph (graphs.GraphedStep): see prepare_capture()
        self._gstep_host = 0             # step() calls + replays so far ...
        self._gstep_dev = None           # ... and the same count on the device once a capture was prepared
        self._capture = None             # the CaptureHandle of the capture in progress
        self._handles = []               # handles with replays not yet folded into state[p]['step']
        self._live = weakref.WeakSet()   # handles of captured steps that are still valid

    META_DTYPE = np.dtype([('p', np.uint64), ('g', np.uint64), ('m', np.uint64), ('v', np.uint64), ('p16', np.uint64),
                           ('numel', np.int64), ('inner', np.int64), ('row_base', np.int64),
                           ('n0', np.int32), ('flags', np.int32), ('step', np.int32), ('reserved', np.int32)])

    def make_master(self, p):
        """Register an fp32 master copy for a parameter that is about to be converted to bf16 (call BEFORE the
        conversion so that no precision is lost)."""
        self.state[p]['master'] = p.detach().to(torch.float32, memory_format=torch.preserve_format).clone()

    FP32_STATE = ('master', 'exp_avg', 'exp_avg_sq')

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict casts every floating-point state tensor to the PARAMETER's dtype, i.e.
        the fp32 master / moments of a bf16 trunk weight would come back as bf16 -- and the kernels, which address them
        as fp32, would write past the buffers.  Restore them from the incoming state dict as fp32, in the parameter's
        memory layout."""
        from itertools import chain
        self._flush_replays()
        self._void_captures()            # step counts change under any captured step
        incoming = {pid: {k: v for k, v in st.items() if k in self.FP32_STATE and torch.is_tensor(v)}
                    for pid, st in state_dict['state'].items()}
        saved_ids = list(chain.from_iterable(g['params'] for g in state_dict['param_groups']))
        super().load_state_dict(state_dict)
        params = list(chain.from_iterable(g['params'] for g in self.param_groups))
        for pid, p in zip(saved_ids, params):
            for k, v in incoming.get(pid, {}).items():
                t = torch.empty_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                t.copy_(v.reshape(p.shape) if v.shape != p.shape else v)
                self.state[p][k] = t
            st = self.state.get(p)
            if st and p.dtype == torch.bfloat16 and 'exp_avg' in st and 'master' not in st:
                # a checkpoint taken from an fp32 model (moments, no master) loaded into a bf16 model: the weight the
                # model holds is the best full-precision value there is
                st['master'] = p.detach().to(torch.float32, memory_format=torch.preserve_format)
        self._plans = {}

    @torch.no_grad()
    def refresh_masters(self, fp32_values=None):
        """After weights were loaded INTO THE MODEL (model.load_state_dict): re-derive the fp32 master of every bf16
        parameter from the loaded weight -- or from `fp32_values` {parameter: fp32 tensor} when the checkpoint carried
        full-precision values -- so that the next step does not overwrite the loaded weights from a stale master."""
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if st is None or p.dtype != torch.bfloat16:
                    continue
                src = fp32_values.get(p) if fp32_values else None
                if 'master' in st:
                    st['master'].copy_(p.detach() if src is None else src)
                elif 'exp_avg' in st:            # state without a master (see load_state_dict): create it
                    st['master'] = (p.detach() if src is None else src.to(p.device)).to(
                        torch.float32, memory_format=torch.preserve_format).clone()
        self._plans = {}

    def master_state_dict(self, model):
        """model.state_dict() with every bf16 trunk weight replaced by its fp32 master (what an apex-O2 checkpoint of
        the reference holds, retrieval_trainer.py:107-111): checkpoints do not lose the low mantissa bits."""
        sd = model.state_dict()
        by_ptr = {p.data_ptr(): p for g in self.param_groups for p in g['params']}
        for k, v in list(sd.items()):
            p = by_ptr.get(v.data_ptr()) if torch.is_tensor(v) else None
            if p is not None and 'master' in self.state.get(p, {}):
                sd[k] = self.state[p]['master'].detach().clone()
        return sd

    @torch.no_grad()
    def broadcast_state(self, src=0, group=None):
        """Multi-rank replicas: make the optimizer state (fp32 masters, both moments, step counts) identical to rank
        `src`'s.  Broadcasting the bf16 weights alone is not enough: the next step rewrites each weight from the
        rank's own master."""
        import torch.distributed as dist
        self._flush_replays()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        self._void_captures()            # the step counts are about to change under any captured step
        # Which parameters carry which state is identical on every rank (same model, same history of None gradients), so the
        # whole state travels as ONE flat fp32 tensor per device plus one int64 vector of step counts -- not ~4 small
        # broadcasts and a host sync per parameter (~2000 collectives for ResNet-101 + BERT-base).
        gloo = dist.get_backend(group) == 'gloo'
        params = [p for g in self.param_groups for p in g['params']]
        if not params:
            return
        dev = params[0].device
        # EVERY rank joins every collective below, whatever state it holds: first rank `src` says which parameters carry which
        # state tensors (one int per parameter), and the others create what they lack (a rank that loaded / resumed nothing
        # yet) or drop what `src` does not have -- a rank returning early while the others broadcast would hang the job.
        sig = torch.tensor([sum(1 << j for j, k in enumerate(self.FP32_STATE) if k in (self.state.get(p) or {})) for p in params],
                           dtype=torch.int64)
        sig = sig if gloo else sig.to(dev)
        dist.broadcast(sig, src, group=group)
        held = []
        for p, bits in zip(params, sig.tolist()):
            if bits == 0:
                self.state.pop(p, None)
                continue
            st = self.state[p]
            for j, k in enumerate(self.FP32_STATE):
                if (bits >> j) & 1 and k not in st:
                    st[k] = (p.detach().to(torch.float32, memory_format=torch.preserve_format).clone() if k == 'master'
                             else torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format))
                elif not (bits >> j) & 1 and k in st:
                    del st[k]
            held.append((p, st))
        if not held:
            return
        tensors = [st[k] for p, st in held for k in self.FP32_STATE if k in st]
        steps = torch.tensor([int(st.get('step', 0)) for p, st in held], dtype=torch.int64)

        def phys(t):
            """(view to read, view to write back into or None): the tensor's bytes as one contiguous run.  channels_last 4-D
            weights are NHWC runs; anything else non-contiguous goes through a contiguous copy and is copied back."""
            if t.is_contiguous():
                return t, None
            if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
                return t.permute(0, 2, 3, 1), None
            return t.contiguous(), t
        chunk, size = [], 0
        for t in tensors + [None]:
            if t is not None and (not chunk or size + t.numel() <= (64 << 20)):      # <= 256 MB of fp32 per collective
                chunk.append(t)
                size += t.numel()
                continue
            if chunk:
                views = [phys(c) for c in chunk]
                flat = torch.cat([v.reshape(-1) for v, _ in views])
                dist.broadcast(flat, src, group=group)
                off = 0
                for c, (v, back) in zip(chunk, views):
                    v.copy_(flat[off:off + c.numel()].view(v.shape))
                    if back is not None:
                        back.copy_(v)
                    off += c.numel()
            chunk, size = ([t], t.numel()) if t is not None else ([], 0)
        steps = steps if gloo else steps.to(dev)
        dist.broadcast(steps, src, group=group)
        for (p, st), n in zip(held, steps.tolist()):
            st['step'] = int(n)
        self._plans = {}

    # ---- steps inside a HIP graph -------------------------------------------------------------------------------------
    # A captured step cannot carry the step count (the bias corrections) or this call's gradient pointers as host values.
    # Inside a stream capture step() therefore (a) uploads the tensor table from a pinned buffer of its own -- the copy is a
    # node of the graph, so every replay restores the CAPTURED gradient addresses (the graph's private pool: static) after
    # whatever an eager step in between uploaded -- and (b) launches cfl_adamp_step_counted, which reads the optimizer's step
    # counter on the device; the table's `step` fields hold every tensor's offset from that counter.  The counter is
    # incremented on the stream by every step (captured `add_`, or eagerly once a capture was prepared).  The host's
    # state[p]['step'] of replayed steps is folded in lazily (CaptureHandle.replayed -> _flush_replays at the next step(),
    # state_dict() or broadcast_state()).  A captured step stays valid while every step in between updates AT LEAST the
    # parameters it updates (its offsets are differences of step counts: a step that skips one of its parameters -- the KD step
    # and the criterion's scalars -- moves the counter without it); CaptureHandle.valid() says so, GraphedStep asks before a replay.

    class CaptureHandle:
        def __init__(self, opt):
            self.opt = opt
            self.plans = []              # the plans of the captured step (kept alive with the graph that addresses their buffers)
            self.params = []             # the parameters it updates
            self.pending = 0             # replays not yet counted in state[p]['step']
            self.nsteps = 0              # step() calls recorded while this capture was open (rolled back if the capture fails)
            self.ids = frozenset()
            self.stale = False

        def replayed(self):
            o = self.opt
            o._gstep_host += 1
            self.pending += 1
            if self not in o._handles:
                o._handles.append(self)
            if len(o._live) > 1:         # another captured step whose parameters this one does not all update
                for h in list(o._live):
                    if h is not self and not h.ids <= self.ids:
                        h.stale = True
                        o._live.discard(h)

        def valid(self):
            return not self.stale

        def __del__(self):               # the graph is gone: its pinned tables go back to their plans
            for plan, pin in self.plans:
                plan['cap_pins'].append(pin)

    def prepare_capture(self, device=None):
        """Call BEFORE the warm-up steps of a step that will be captured: creates the device-side step counter (an
        allocation + a fill cannot happen inside the capture: they would be replayed)."""
        if self._gstep_dev is None:
            if device is None:
                device = next(p.device for g in self.param_groups for p in g['params'])
            self._gstep_dev = torch.tensor([self._gstep_host], dtype=torch.int32, device=device)

    def capture_begin(self):
        if self._gstep_dev is None:
            raise _lib.CreamflHipError('AdamP.capture_begin() without prepare_capture()')
        self._flush_replays()
        self._capture = AdamP.CaptureHandle(self)
        return self._capture

    def capture_end(self, handle, ok=True):
        """ok=False: the capture FAILED after step() had been recorded (an exception inside the captured function, or from
        hipStreamEndCapture).  Nothing that was recorded ever ran -- no update, no captured `add_` on the device counter -- but the
        host has counted the step: state[p]['step'] and the host's running count are taken back, so that the eager re-run of the
        same step is counted once and later graphs of this optimizer compute their offsets from a count that equals the device's."""
        if self._capture is handle:
            self._capture = None
        if not ok:
            for p in handle.params:                  # (a parameter stepped twice inside the capture is listed twice)
                st = self.state.get(p)
                if st is not None and 'step' in st:
                    st['step'] -= 1
            self._gstep_host -= handle.nsteps
            for plan, pin in handle.plans:           # the pinned tables go back now; the device table is re-uploaded by the next step
                plan['cap_pins'].append(pin)
            handle.plans, handle.params, handle.nsteps = [], [], 0
            handle.stale = True
            return handle
        handle.ids = frozenset(id(p) for p in handle.params)
        self._live.add(handle)
        return handle

    def _void_captures(self):
        for h in list(self._live):
            h.stale = True
        self._live.clear()

    def _flush_replays(self):
        for h in self._handles:
            if h.pending:
                for p in h.params:
                    self.state[p]['step'] += h.pending
                h.pending = 0
        del self._handles[:]

    def state_dict(self):
        self._flush_replays()
        return super().state_dict()

    PIN_SLOTS = 4

    @staticmethod
    def _upload_meta(plan):
        slot = plan['pins'][plan['pin_next']]
        plan['pin_next'] = (plan['pin_next'] + 1) % len(plan['pins'])
        capturing = torch.cuda.is_current_stream_capturing()
        if slot['event'] is not None and not capturing:
            slot['event'].synchronize()          # the previous upload from this slot has been consumed
        slot['view'][:] = plan['meta']
        plan['meta_dev'].copy_(slot['pin'], non_blocking=True)
        if not capturing:
            slot['event'] = torch.cuda.Event()
            slot['event'].record()

    def _plan(self, gi, params, clip_ids):
        key = (gi, tuple((p.data_ptr(), p.dtype) for p in params), tuple(sorted(clip_ids)) if clip_ids else ())
        plan = self._plans.get(gi)
        if plan is not None and plan['key'] == key:
            return plan
        dev = params[0].device
        meta = np.zeros(len(params), dtype=self.META_DTYPE)
        items, matrix_ids = [], []
        row_base = 0
        for t, p in enumerate(params):
            numel = p.numel()
            if p.dim() > 1:
                n0 = p.shape[0]
                inner = numel // n0
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not dense or p.stride(0) != inner:
                    raise _lib.CreamflHipError(f'AdamP: parameter {tuple(p.shape)} is not dense with dim 0 outermost')
                flags = 1
                rows_per = max(1, self.ROWS_TARGET // max(inner, 1))
                for r0 in range(0, n0, rows_per):
                    items.append((t, r0, min(rows_per, n0 - r0)))
                matrix_ids.append(t)
                meta[t]['row_base'] = row_base
                row_base += n0
            else:
                n0, inner, flags = 1, numel, 0
                for e0 in range(0, numel, self.ROWS_TARGET):
                    items.append((t, e0, min(self.ROWS_TARGET, numel - e0)))
            if id(p) in clip_ids:
                flags |= 2
            st = self.state[p]
            for k in self.FP32_STATE:                # the kernels address these as fp32 in the parameter's layout
                if k in st and (st[k].dtype != torch.float32 or st[k].stride() != p.stride() or st[k].device != p.device):
                    raise _lib.CreamflHipError(f'AdamP: state[{k!r}] of a {tuple(p.shape)} parameter is {st[k].dtype} / strides '
                                               f'{st[k].stride()} (need fp32 / {p.stride()}): was it loaded through '
                                               f'torch.optim.Optimizer.load_state_dict instead of AdamP.load_state_dict?')
            if p.dtype == torch.bfloat16:            # bf16 model weight, fp32 master (apex-O2 style)
                flags |= 4
                meta[t]['p'] = st['master'].data_ptr(); meta[t]['p16'] = p.data_ptr()
            else:
                meta[t]['p'] = p.data_ptr()
            meta[t]['m'] = st['exp_avg'].data_ptr(); meta[t]['v'] = st['exp_avg_sq'].data_ptr()
            meta[t]['numel'] = numel; meta[t]['inner'] = inner; meta[t]['n0'] = n0; meta[t]['flags'] = flags
        # Uploads go through a ring of pinned staging buffers (async, graph-capturable).  The host runs up to a
        # step ahead of the GPU, so a slot is rewritten only after the event recorded behind its last upload.
        pins = []
        for _ in range(self.PIN_SLOTS):
            t_pin = torch.empty(meta.nbytes, dtype=torch.uint8).pin_memory()
            pins.append({'pin': t_pin, 'view': t_pin.numpy().view(self.META_DTYPE), 'event': None})
        plan = {
            'key': key, 'meta': meta, 'pins': pins, 'pin_next': 0,
            'meta_dev': torch.empty(meta.nbytes, dtype=torch.uint8, device=dev),
            'items': torch.tensor(items, dtype=torch.int32, device=dev).reshape(-1, 3).contiguous(),
            'matrix_ids': torch.tensor(matrix_ids or [0], dtype=torch.int32, device=dev),
            'n_matrix': len(matrix_ids),
            'rowstats': torch.empty(max(row_base, 1) * 4, dtype=torch.float32, device=dev),
            'tstats': torch.ones(len(params), dtype=torch.float32, device=dev),
            'partial': torch.empty(len(items), dtype=torch.float32, device=dev),
            'clip': torch.ones(2, dtype=torch.float32, device=dev),
            'gptrs': None,
            'cap_pins': [torch.empty(meta.nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)],
        }
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """clip = (iterable of parameters, max_norm): fuse nn.utils.clip_grad_norm_ over those parameters into
        the step (the reference clips model parameters only, retrieval_trainer.py:211-213).  Returns the loss
        of `closure` (None by default)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        clip_ids, max_norm = (set(), 0.0)
        if clip is not None:
            clip_ids, max_norm = {id(p) for p in clip[0]}, float(clip[1])
        self.last_grad_norm = None
        capturing = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
        if capturing and self._capture is None:
            raise _lib.CreamflHipError('AdamP.step() inside a stream capture needs prepare_capture() before the warm-up steps and '
                                       'capture_begin() / capture_end() around the capture (graphs.GraphedStep(optimizer=...))')
        if self._handles:
            self._flush_replays()
        self._gstep_host += 1
        if capturing:
            self._capture.nsteps += 1        # (taken back by capture_end(handle, ok=False) if the capture fails)
        if self._gstep_dev is not None:
            self._gstep_dev.add_(1)          # (a node of the graph when capturing)
        stepped = set()
        if self.grad_override is not None and self.grad_override_consume is not None:
            self.grad_override_consume()
        work = []
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            if not all(p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) for p in params):
                raise _lib.CreamflHipError('creamfl_amd AdamP needs fp32 / bf16 parameters on a HIP device (no CPU path)')
            for p in params:
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    if p.dtype == torch.bfloat16 and 'master' not in st:
                        st['master'] = p.detach().to(torch.float32, memory_format=torch.preserve_format)
                st['step'] += 1
            if self._live:
                stepped.update(id(p) for p in params)
            plan = self._plan(gi, params, clip_ids)
            grads = []
            ov = self.grad_override
            for p in params:
                g = p.grad if ov is None else ov.get(p, p.grad)
                if g.dtype != p.dtype or g.stride() != p.stride():
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    g = g2
                grads.append(g)
            gptrs = [g.data_ptr() for g in grads]
            # adamp.AdamP keeps `step` per parameter; parameters whose gradient was None in some steps (the criterion's
            # scalars during the KD phase, a whole tower when only one kind of client exists) lag behind.  Uniform steps
            # travel as the launch argument; otherwise every tensor's own count goes into its meta record.
            steps = [int(self.state[p]['step']) for p in params]
            tsteps = np.zeros(len(params), dtype=np.int32) if min(steps) == max(steps) else np.asarray(steps, dtype=np.int32)
            if capturing:
                # the table of THIS graph: captured gradient addresses, step offsets from the device counter; uploaded by a node
                # of the graph from a pinned buffer nothing else writes
                cap = plan['meta'].copy()
                cap['g'] = np.asarray(gptrs, dtype=np.uint64)
                cap['step'] = np.asarray(steps, dtype=np.int64) - self._gstep_host
                if not plan['cap_pins']:
                    raise _lib.CreamflHipError('AdamP: no pinned table left for another capture of this step (two live graphs '
                                               'already address it; drop the old GraphedStep first)')
                pin = plan['cap_pins'].pop()     # (pinned memory cannot be allocated inside a capture: a pool made with the plan)
                pin.numpy().view(self.META_DTYPE)[:] = cap
                plan['meta_dev'].copy_(pin, non_blocking=True)
                plan['captured'] = True          # the device table is the graph's after every replay: eager steps re-upload
                self._capture.plans.append((plan, pin))
                self._capture.params.extend(params)
            elif plan.get('captured') or gptrs != plan['gptrs'] or not np.array_equal(plan['meta']['step'], tsteps):
                plan['meta']['g'] = np.asarray(gptrs, dtype=np.uint64)
                plan['meta']['step'] = tsteps
                self._upload_meta(plan)
                plan['gptrs'] = gptrs
            stream = ctypes.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream)
            n_items = plan['items'].shape[0]
            clipped = bool(clip_ids and max_norm > 0 and any(id(p) in clip_ids for p in params))
            if clipped:
                _lib.check(lib.cfl_grad_clip_coef(plan['meta_dev'].data_ptr(), plan['items'].data_ptr(), n_items, max_norm,
                                                  plan['partial'].data_ptr(), plan['clip'].data_ptr(), stream),
                           'cfl_grad_clip_coef')
            work.append((group, params, plan, grads, steps, stream, n_items, clipped))
        for h in list(self._live):           # (an eager step between replays, or another captured step being recorded)
            if not h.ids <= stepped:
                h.stale = True
                self._live.discard(h)
        clipped_plans = [w[2] for w in work if w[7]]
        if len(clipped_plans) > 1:
            # clip_grad_norm_ is ONE norm over every clipped parameter, whatever group it sits in: combine the groups' norms
            # on the device (still no host synchronisation) and hand every group the same coefficient
            total = torch.stack([pl['clip'][0] for pl in clipped_plans]).square().sum().sqrt()
            coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
            both = torch.stack([total, coef])
            for pl in clipped_plans:
                pl['clip'].copy_(both)
        if clipped_plans:
            self.last_grad_norm = clipped_plans[0]['clip'][0]
        for group, params, plan, grads, steps, stream, n_items, clipped in work:
            clip_ptr = ctypes.c_void_p(plan['clip'].data_ptr()) if clipped else ctypes.c_void_p(0)
            beta1, beta2 = group['betas']
            if capturing:
                _lib.check(lib.cfl_adamp_step_counted(
                    plan['meta_dev'].data_ptr(), len(params), plan['items'].data_ptr(), n_items, plan['matrix_ids'].data_ptr(),
                    plan['n_matrix'], plan['rowstats'].data_ptr(), plan['tstats'].data_ptr(), float(group['lr']), float(beta1),
                    float(beta2), float(group['eps']), float(group['weight_decay']), float(group['delta']),
                    float(group['wd_ratio']), int(bool(group['nesterov'])), self._gstep_dev.data_ptr(), clip_ptr, stream),
                    'cfl_adamp_step_counted')
                continue
            _lib.check(lib.cfl_adamp_step(plan['meta_dev'].data_ptr(), len(params), plan['items'].data_ptr(), n_items,
                                          plan['matrix_ids'].data_ptr(), plan['n_matrix'], plan['rowstats'].data_ptr(),
                                          plan['tstats'].data_ptr(), float(group['lr']), float(beta1), float(beta2),
                                          float(group['eps']), float(group['weight_decay']), float(group['delta']),
                                          float(group['wd_ratio']), int(bool(group['nesterov'])),
                                          max(steps), clip_ptr, stream), 'cfl_adamp_step')
        del work
        return loss


def get_optimizer(optimizer_name, parameters, config, logger=None):
    if logger:
        logger.log('creating [{}] from Config({})'.format(optimizer_name, config))
    if optimizer_name == 'adam':
        if set(config.keys()) - {'learning_rate', 'betas', 'eps', 'weight_decay', 'amsgrad', 'name'}:
            warnings.warn('found unused keys in {}'.format(config.keys()))
        optimizer = optim.Adam(parameters, lr=config.learning_rate, betas=config.get('betas', (0.9, 0.999)),
                               eps=float(config.get('eps', 1e-8)), weight_decay=float(config.get('weight_decay', 0)),
                               amsgrad=config.get('amsgrad', False))
    elif optimizer_name == 'adamn' or optimizer_name == 'adamp':
        if set(config.keys()) - {'learning_rate', 'betas', 'eps', 'weight_decay', 'name'}:
            warnings.warn('found unused keys in {}'.format(config.keys()))
        optimizer = AdamP(parameters, lr=config.learning_rate, betas=config.get('betas', (0.9, 0.999)),
                          eps=float(config.get('eps', 1e-8)), weight_decay=float(config.get('weight_decay', 0)))
    else:
        raise ValueError(f'Invalid optimizer name: {optimizer_name}')
    return optimizer


def get_lr_scheduler(scheduler_name, optimizer, config, logger=None):
    if logger:
        logger.log('creating [{}] from Config({})'.format(scheduler_name, config))
    if scheduler_name == 'reduce_lr_on_plateau':
        if set(config.keys()) - {'mode', 'factor', 'patience', 'verbose', 'threshold', 'threshold_mode', 'cooldown',
                                 'min_lr', 'eps', 'name'}:
            warnings.warn('found unused keys in {}'.format(config.keys()))
        lr_scheduler = optim.lr_scheduler.ReduceLROnPlateau(
            optimizer, mode=config.get('mode', 'min'), factor=float(config.get('factor', 0.1)),
            patience=config.get('patience', 10), threshold=float(config.get('threshold', 1e-4)),
            threshold_mode=config.get('threshold_mode', 'rel'), cooldown=float(config.get('cooldown', 0)),
            min_lr=float(config.get('min_lr', 0)), eps=float(config.get('eps', 1e-8)))
    elif scheduler_name == 'cosine_annealing':
        lr_scheduler = optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config.T_max)
    else:
        raise ValueError(f'Invalid scheduler name: {scheduler_name}')
    return lr_scheduler
