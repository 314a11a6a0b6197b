"""Flat parameter / gradient arenas, bucketed gradient exchange and the HIP optimizer step.

MI355X-first layout for the data-parallel training step (SURVEY.md 8e "Training", 8f rank 4):

* every trainable parameter of the model is a VIEW into one flat fp32 arena (`flat_p`); its gradient is a view into
  a second arena of the same shape (`flat_g`).  288 GB of HBM makes the extra copies (momentum, EMA) free, and a flat
  arena turns the optimizer + EMA + gradient reset into ONE HBM-bound kernel launch (`rart_sgd_step_f32` /
  `rart_adamw_step_f32`) instead of ~160 per-tensor launches;
* the gradient exchange is a handful of large all-reduces on contiguous slices of `flat_g` ("buckets", default
  48 MiB: xGMI rings are per-link bound, so few large messages beat many small ones), launched asynchronously from
  post-accumulate-grad hooks as soon as the last gradient of a bucket has been produced, i.e. overlapped with the
  rest of the backward pass (`dist.sync: False` semantics of the reference configs); the 1/world_size of the mean
  is folded into the optimizer kernel's `grad_scale`, so no separate scaling pass exists;
* parameters that the reference configs exclude from weight decay would sit in their own contiguous range;
  `no_decay(name, p)` chooses the range.

The arithmetic is torch.optim.SGD / AdamW (exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:11-33,
new_adv_train/vit_base/config.yaml:11-38); the kernels are pinned to it through oracle/train_ref.py.
"""
import os
import time

import torch
import torch.distributed as dist


class ParamArena:
    def __init__(self, model, bucket_bytes=48 << 20, no_decay=None, overlap=True):
        params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not params:
            raise ValueError('ParamArena: the model has no trainable parameters')
        device = params[0][1].device
        no_decay = no_decay or (lambda name, p: False)
        # decayed parameters first, then the no-decay range; registration order inside each range
        order = [x for x in params if not no_decay(*x)] + [x for x in params if no_decay(*x)]
        self.n_decay = sum(p.numel() for n, p in order if not no_decay(n, p))
        self.names = [n for n, _ in order]
        self.params = [p for _, p in order]
        sizes = [p.numel() for p in self.params]
        # every parameter starts on a 16-byte boundary so the vectorised kernels and RCCL see aligned slices
        self.offsets, off = [], 0
        for s in sizes:
            self.offsets.append(off)
            off += (s + 3) // 4 * 4
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        for p, o in zip(self.params, self.offsets):
            v = self.flat_p[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)
        k = len([1 for n, p in order if not no_decay(n, p)])
        self.decay_end = self.offsets[k] if k < len(self.params) else off
        # buckets: contiguous element ranges, filled in REVERSE parameter order (backward produces gradients
        # roughly last-layer first)
        self.buckets = []            # [lo, hi, n_params]
        self.bucket_of = {}
        cap = max(bucket_bytes // 4, 1)
        hi = off
        cur = []
        for i in reversed(range(len(self.params))):
            cur.append(i)
            lo = self.offsets[i]
            if hi - lo >= cap or i == 0:
                b = len(self.buckets)
                self.buckets.append([lo, hi, len(cur)])
                for j in cur:
                    self.bucket_of[j] = b
                cur, hi = [], lo
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the exchange runs whenever there is more than one rank -- and, with RART_FORCE_DIST=1, also in a one-rank process group, so that a
        # single-GPU box executes the very same bucketed all_reduce calls on arena slices (tests/test_rccl_gpu.py)
        self.exchange = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get('RART_FORCE_DIST') == '1')
        self._hooks = []
        self.overlap = overlap
        self._index = {id(p): i for i, p in enumerate(self.params)}
        if self.exchange and overlap:      # dist.sync: True -> every bucket is reduced after backward instead
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ---- gradient exchange -------------------------------------------------------------------------------------
    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._pending[b] += 1
            if self._pending[b] == self.buckets[b][2]:
                lo, hi, _ = self.buckets[b]
                self._handles.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        return hook

    def grad_ready(self, param):
        """Same bookkeeping as the autograd hook, for gradients produced outside autograd (the HIP train engine)."""
        if self.exchange and self.overlap:
            i = self._index.get(id(param))
            if i is not None:
                self._make_hook(i)(param)

    def finish_grad_exchange(self):
        """Wait for the bucket all-reduces launched during backward; buckets whose parameters produced no gradient
        this step (unused branches) are reduced here so every rank issues the same collectives.  Returns the
        factor the optimizer must fold into the gradients (1 / world_size)."""
        if self.exchange:
            late = 0
            for b, (lo, hi, cnt) in enumerate(self.buckets):
                if self._pending[b] != cnt:
                    self._handles.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
                    late += 1
            ev = None
            if self.flat_g.is_cuda:
                # Work.wait() of an RCCL collective makes the compute stream wait for the communication stream without blocking the
                # host: the stall the exchange really costs is the time between these two events on the compute stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t0 = time.perf_counter()
            for h in self._handles:
                h.wait()
            host_wait = time.perf_counter() - t0
            if ev is not None:
                ev[1].record()
            self._last_exchange = dict(buckets=len(self.buckets), launched_during_backward=len(self.buckets) - late, launched_after_backward=late,
                                       bytes=4 * int(self.flat_g.numel()), bucket_bytes=[4 * (hi - lo) for lo, hi, _ in self.buckets],
                                       host_wait_s=host_wait, world=self.world, _events=ev)
            self._handles = []
            self._pending = [0] * len(self.buckets)
        return 1.0 / self.world

    def exchange_stats(self):
        """What the last finish_grad_exchange() did -- bucket count and bytes, how many buckets started during backward (overlapped) and how
        many after it, the host-side wait and the stall of the compute stream behind the collectives (synchronises on the closing event:
        call it at logging frequency, not every step).  None when there is no exchange."""
        st = getattr(self, '_last_exchange', None)
        if st is None:
            return None
        st = dict(st)
        ev = st.pop('_events')
        if ev is not None:
            ev[1].synchronize()
            st['stream_wait_s'] = ev[0].elapsed_time(ev[1]) * 1e-3
        return st

    def repoint_grads(self):
        """autograd replaces `.grad` when it was None; keep the arena views (call after zero_grad(set_to_none=True))."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)


class HipOptimizer:
    """SGD-Nesterov / AdamW + EMA + gradient reset as one HIP launch per weight-decay range of a ParamArena."""

    def __init__(self, arena, kind='SGD', lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, ema_decay=None):
        from .. import _lib
        if arena.flat_p.device.type != 'cuda':
            raise RuntimeError('HipOptimizer needs the parameters on an MI355X (no CPU fallback)')
        self.lib = _lib.load()
        self._lib_mod = _lib
        self.arena, self.kind = arena, kind
        self.lr, self.momentum, self.nesterov, self.weight_decay = lr, momentum, nesterov, weight_decay
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.m = torch.zeros_like(arena.flat_p)
        self.v = torch.zeros_like(arena.flat_p) if kind == 'AdamW' else None
        self.ema_decay = ema_decay
        self.ema = arena.flat_p.clone() if ema_decay is not None else None

    def _ranges(self):
        a = self.arena
        if a.decay_end >= a.numel:
            return [(0, a.numel, self.weight_decay)]
        return [(0, a.decay_end, self.weight_decay), (a.decay_end, a.numel, 0.0)]

    def step(self, grad_scale=1.0):
        a, lib = self.arena, self.lib
        self.step_count += 1
        st = self._lib_mod.stream_ptr()
        for lo, hi, wd in self._ranges():
            n = hi - lo
            if n <= 0:
                continue
            p, g, m = (t.data_ptr() + 4 * lo for t in (a.flat_p, a.flat_g, self.m))
            e = self.ema.data_ptr() + 4 * lo if self.ema is not None else None
            dec = self.ema_decay if self.ema is not None else 0.0
            if self.kind == 'AdamW':
                self._lib_mod.check(lib.rart_adamw_step_f32(p, g, m, self.v.data_ptr() + 4 * lo, e, n, self.lr,
                                                             self.betas[0], self.betas[1], self.eps, wd,
                                                             self.step_count, grad_scale, dec, 1, st))
            else:
                self._lib_mod.check(lib.rart_sgd_step_f32(p, g, m, e, n, self.lr, self.momentum, wd,
                                                           1 if self.nesterov else 0, grad_scale, dec, 1, st))

    def state_dict(self):
        """Everything a resumed run needs besides the parameters: momentum / first moment, Adam's second moment, the step
        count (Adam's bias correction) -- tensors and numbers only, so the checkpoint loads with weights_only=True.  The EMA
        travels under the checkpoint's own 'ema' key (module names)."""
        sd = {'kind': self.kind, 'step_count': int(self.step_count), 'm': self.m.detach().cpu()}
        if self.v is not None:
            sd['v'] = self.v.detach().cpu()
        return sd

    def load_state_dict(self, sd):
        if sd.get('kind', self.kind) != self.kind or sd['m'].numel() != self.m.numel():
            raise ValueError('HipOptimizer.load_state_dict: the checkpoint belongs to another optimizer / parameter arena')
        self.step_count = int(sd.get('step_count', 0))
        self.m.copy_(sd['m'].to(self.m.device))
        if self.v is not None and sd.get('v') is not None:
            self.v.copy_(sd['v'].to(self.v.device))

    def ema_state_dict(self, model):
        """EMA parameters under the module's names (floating-point buffers are tracked by the solver)."""
        out = {}
        for name, p, o in zip(self.arena.names, self.arena.params, self.arena.offsets):
            out[name] = self.ema[o:o + p.numel()].view_as(p).clone()
        return out


def label_smooth_ce(logits, labels, smoothing, scale):
    """(loss_rows, dlogits) of F.cross_entropy(label_smoothing) through rart_label_smooth_ce_f32."""
    from .. import _lib
    lib = _lib.load()
    logits = logits.float().contiguous()
    b, c = logits.shape
    loss = torch.empty(b, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    _lib.check(lib.rart_label_smooth_ce_f32(logits.data_ptr(), labels.contiguous().data_ptr(), b, c, smoothing, scale,
                                            loss.data_ptr(), dl.data_ptr(), _lib.stream_ptr()))
    return loss, dl
