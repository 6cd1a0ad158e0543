"""`optim.Adam` of the train scripts (train_model.py:273-274, train_feature_creation_gen.py:157-160) as ONE HIP launch per
step for all parameter tensors of a module, instead of torch's ~7 multi-tensor launches.

Same constructor arguments, `param_groups` keys and per-parameter state (`step`, `exp_avg`, `exp_avg_sq`) as
`torch.optim.Adam`, so `state_dict()` / `load_state_dict()` interoperate with it in both directions.  Only what the
reference uses is implemented on the GPU path (no weight decay, no amsgrad, no maximize); asking for anything else raises.
"""
import ctypes

import torch

from . import _lib, ops

L = _lib.lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 grad_scale=1.0, zero_grads=False, capturable=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise NotImplementedError("weight_decay / amsgrad / maximize are not used by the reference and have no HIP kernel")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        # grad is read as grad * grad_scale (e.g. 1/N of a summed data-parallel gradient); zero_grads: the kernel leaves
        # zeros in .grad (for callers that keep persistent gradient buffers instead of zero_grad(set_to_none=True))
        self.grad_scale = float(grad_scale)
        self.zero_grads = bool(zero_grads)
        # capturable: the step-dependent scalars live in device memory, so a captured HIP graph of step() stays valid from
        # one replay to the next; the owner of the graph calls advance() before every replay (see loop.py)
        self.capturable = bool(capturable)
        self._dyn = {}
        self._tables = {}
        self._pack_tables = {}

    def _table(self, key, entries):
        """ctypes tensor table for one (group, step) cohort; param / state pointers are stable, gradients are re-read."""
        hit = self._tables.get(key)
        n = len(entries)
        if hit is None or hit[1] != n:
            arr = (_lib.AdamTensor * n)()
            hit = (arr, n)
            self._tables[key] = hit
        arr = hit[0]
        for i, (p, g, m, v) in enumerate(entries):
            t = arr[i]
            t.param = p.data_ptr(); t.grad = g.data_ptr(); t.exp_avg = m.data_ptr(); t.exp_avg_sq = v.data_ptr()
            t.numel = p.numel()
        return arr

    def _cohort(self, group):
        """(entries, step tensors) of the parameters of `group` that have a gradient; creates missing state."""
        entries, steps = [], []
        for p in group['params']:
            g = p.grad
            if g is None:
                continue
            if g.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients")
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise _lib.SempyrError("optim.Adam: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape):
                g = g.to(device=p.device, dtype=torch.float32).contiguous()
                p.grad = g
            st = self.state[p]
            if len(st) == 0:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.SempyrError("optim.Adam: optimizer state must exist before step() is captured into a graph "
                                           "(run one eager step first)")
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.append(st['step'])
            entries.append((p, g, st['exp_avg'], st['exp_avg_sq']))
        return entries, steps

    @torch.no_grad()
    def advance(self):
        """capturable mode: count one step for every parameter and hand the step scalars {-lr/(1-b1^t), sqrt(1-b2^t)} to
        the device (stream-ordered on the current stream).  step() calls it itself in eager execution; the owner of a
        captured graph calls it before each replay."""
        for gi, group in enumerate(self.param_groups):
            steps = [self.state[p]['step'] for p in group['params'] if len(self.state[p]) != 0]
            if not steps:
                continue
            t = int(steps[0]) + 1
            torch._foreach_add_(steps, 1)
            beta1, beta2 = group['betas']
            out = (ctypes.c_float * 2)()
            _lib.check(L.sempyr_adam_scalars(t, float(group['lr']), float(beta1), float(beta2), out), "sempyr_adam_scalars")
            dyn = self._dyn_buffer(gi, group)
            words = (ctypes.c_uint32 * 2).from_buffer(out)
            ops.store_words(dyn, [int(words[0]), int(words[1])])

    def _dyn_buffer(self, gi, group):
        dyn = self._dyn.get(gi)
        if dyn is None:
            dev = group['params'][0].device
            dyn = torch.zeros(2, dtype=torch.float32, device=dev)
            self._dyn[gi] = dyn
        return dyn

    @torch.no_grad()
    def _step_dev(self):
        capturing = torch.cuda.is_current_stream_capturing()
        for gi, group in enumerate(self.param_groups):
            if group['weight_decay'] != 0 or group['amsgrad'] or group.get('maximize', False):
                raise NotImplementedError("weight_decay / amsgrad / maximize have no HIP kernel")
            entries, steps = self._cohort(group)
            if not entries:
                continue
            if len(entries) != sum(1 for p in group['params'] if len(self.state[p]) != 0):
                raise _lib.SempyrError("optim.Adam(capturable=True): every parameter with optimizer state needs a gradient in "
                                       "every step (they share one step count)")
        if not capturing:
            self.advance()
        for gi, group in enumerate(self.param_groups):
            entries, steps = self._cohort(group)
            if not entries:
                continue
            beta1, beta2 = group['betas']
            dev = entries[0][0].device
            arr = self._table((gi, 'dev'), entries)
            stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
            with torch.cuda.device(dev):
                _lib.check(L.sempyr_adam_step_dev(arr, len(entries), self._dyn_buffer(gi, group).data_ptr(), float(beta1),
                                                  float(beta2), float(group['eps']), self.grad_scale,
                                                  1 if self.zero_grads else 0, stream), "sempyr_adam_step_dev")
            torch.autograd.graph.increment_version([e[0] for e in entries])
            ops.repack_cached(group['params'], self._pack_tables.setdefault(gi, {}))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            self._step_dev()
            return loss
        for gi, group in enumerate(self.param_groups):
            if group['weight_decay'] != 0 or group['amsgrad'] or group.get('maximize', False):
                raise NotImplementedError("weight_decay / amsgrad / maximize have no HIP kernel")
            beta1, beta2 = group['betas']
            cohorts = {}                       # step count -> tensors (one cohort unless some parameter skipped steps)
            steps = []
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.SempyrError("optim.Adam: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
                if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape):
                    g = g.to(device=p.device, dtype=torch.float32).contiguous()
                    p.grad = g
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.append(st['step'])
                cohorts.setdefault(int(st['step']) + 1, []).append((p, g, st['exp_avg'], st['exp_avg_sq']))
            if not steps:
                continue
            torch._foreach_add_(steps, 1)
            for t, entries in cohorts.items():
                dev = entries[0][0].device
                if any(e[0].device != dev for e in entries):
                    raise _lib.SempyrError("optim.Adam: the parameters of one group must live on one GPU")
                arr = self._table((gi, len(cohorts) > 1 and t), entries)
                stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
                with torch.cuda.device(dev):
                    _lib.check(L.sempyr_adam_step(arr, len(entries), t, float(group['lr']), float(beta1), float(beta2),
                                                  float(group['eps']), self.grad_scale, 1 if self.zero_grads else 0, stream),
                               "sempyr_adam_step")
                # the kernel wrote through raw pointers: bump the autograd version counters (packed-weight and
                # eval-BatchNorm caches are keyed on them)
                torch.autograd.graph.increment_version([e[0] for e in entries])
            # the convolutions read weights through packed copies: refresh all of them with one launch
            ops.repack_cached(group['params'], self._pack_tables.setdefault(gi, {}))
        return loss
