"""Data parallelism: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on ROCm; "gloo"
for the CPU tests), identical parameters on every rank and ONE all-reduce of a flat fp32 gradient
buffer per optimizer step.  The reference has no distributed code at all; this is
the only collective on the path.  BatchNorm statistics and the diversity ratio stay per replica.

The flat buffer is a persistent tensor per module (7.0 M floats for D, 9.5 M for G).  With
`flatten_grads(module)` every `param.grad` IS a view of it: the weight-gradient kernels and the
gradient-accumulation launch add into it in place, the collective runs on it directly (xGMI links are
point-to-point, so one 28-38 MB exchange beats ~100 small ones), `optim.Adam(grad_scale=1/N,
zero_grads=True)` reads the sum as a mean and leaves zeros behind for the next step -- no pack, no scale,
no unpack pass.  Modules without flat gradients take the pack / reduce / scale / unpack route.
"""
import os

import torch
import torch.distributed as dist

_FLAT = {}
# critic gradient buckets exchanged from inside the backward pass (FlatGrads); 0 = one collective after the pass
BUCKET_BYTES = int(os.environ.get('SEMPYR_DP_BUCKET_BYTES', str(8 << 20)))


def rank_from_env():
    return int(os.environ.get('RANK', '0'))


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def init_device(backend=None):
    """Initialise the process group from the torchrun environment (if any) and pick this rank's GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("the sempyr hot path needs a GPU (no CPU fallback)")
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or os.environ.get('SEMPYR_DIST_BACKEND', 'nccl')
    if os.environ.get('SEMPYR_SHARE_GPU', '0') == '1':
        # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device,
        # so this needs SEMPYR_DIST_BACKEND=gloo); never used for measurements
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return torch.device('cuda', local)


def _host_staged():
    """gloo moves device tensors through the host anyway; staging them explicitly keeps its CUDA path (which hangs when two
    rehearsal ranks share one GPU) out of the picture.  RCCL works on device memory directly."""
    return dist.get_backend() == 'gloo'


def broadcast_module(module, src=0):
    """Make parameters and buffers identical on all ranks (rank `src` wins): one flat collective per dtype."""
    if not is_distributed():
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for tensors in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        if flat.is_cuda and _host_staged():
            host = flat.cpu()
            dist.broadcast(host, src)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def _flat_buffer(params):
    # keyed by size, not by tensor identity: gradient tensors are new allocations every step
    n = sum(p.numel() for p in params)
    key = (n, params[0].dtype, params[0].device)
    buf = _FLAT.get(key)
    if buf is None:
        buf = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        _FLAT[key] = buf
    return buf


def allreduce_mean_(tensors, group=None):
    """In-place mean over ranks of a list of same-dtype tensors through one flat collective."""
    if not tensors:
        return
    ws = dist.get_world_size(group)
    flat = _flat_buffer(tensors)
    views = []
    off = 0
    for t in tensors:
        views.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    torch._foreach_copy_(views, tensors)
    if flat.is_cuda and _host_staged():
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / ws)
    torch._foreach_copy_(tensors, views)


class _Bucket:
    __slots__ = ("start", "end", "ids", "left")

    def __init__(self, start, end, ids):
        self.start, self.end, self.ids, self.left = start, end, ids, 0


class FlatGrads:
    """Persistent flat fp32 gradient buffer of a module; every `param.grad` is a view of it (see the module docstring).

    Layout: the weight tensors (dim >= 2) in REVERSE registration order -- the order in which a backward pass finishes
    them -- followed by the 1-D tensors (biases, norm affine parameters).  With `bucket_bytes` > 0 the leading weights are
    grouped into buckets of at least that size; a bucket is all-reduced from inside the backward pass, on the side stream
    the weight-gradient kernels run on, as soon as the last gradient contribution of its last weight has been issued
    (`ops._GRAD_ISSUED_HOOK`), so the exchange runs over xGMI under the rest of the backward pass.  Everything behind the
    last bucket (small early-layer weights and every 1-D tensor, whose gradients are only complete when the pass ends) is
    one more collective in `sync_gradients`.  How many contributions a weight receives per step (three in a critic step:
    logit pass, and both chains of the gradient penalty's double backward) is LEARNED: buckets fire early only after two
    consecutive steps with identical counts; until then, and whenever a step's counts fall short, they fire in
    `sync_gradients`.  A contribution that arrives after its bucket has been reduced raises (it would be lost otherwise).
    The collectives are always issued in the same order (buckets in layout order, then the tail) on every rank."""

    def __init__(self, module, bucket_bytes=0):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params or any(p.dtype != torch.float32 or not p.is_contiguous() for p in params):
            raise ValueError("flatten_grads: contiguous fp32 parameters expected")
        dev = params[0].device
        weights = [p for p in params if p.dim() >= 2]
        order = weights[::-1] + [p for p in params if p.dim() < 2]
        # every view starts on a 16-byte boundary (float4 accesses of the Adam / accumulate kernels)
        offs, n = {}, 0
        for p in order:
            offs[id(p)] = n
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params, self.views = params, []
        for p in params:
            off = offs[id(p)]
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
        self.buckets, self.bucket_of = [], {}
        if bucket_bytes > 0:
            start, ids = 0, []
            for k, p in enumerate(weights[::-1]):
                ids.append(id(p))
                end = offs[id(weights[::-1][k + 1])] if k + 1 < len(weights) else None
                if end is not None and 4 * (end - start) >= bucket_bytes:
                    b = _Bucket(start, end, ids)
                    for i in ids:
                        self.bucket_of[i] = b
                    self.buckets.append(b)
                    start, ids = end, []
        self.tail_start = self.buckets[-1].end if self.buckets else 0
        self.expected, self.stable, self.armed = None, 0, False
        self.seen, self.fired, self.works = {}, 0, []
        self.stats = {"early": 0, "late": 0, "steps": 0}

    def reattach(self):
        """put the views back after something replaced `.grad` (zero_grad(set_to_none=True), a foreign optimizer)"""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    # ---- bucketed exchange ------------------------------------------------------------------------------
    def begin_step(self):
        self.seen, self.fired, self.works = {}, 0, []
        for b in self.buckets:
            b.left = len(b.ids)

    def _reduce_range(self, start, end, asynchronous):
        view = self.flat[start:end]
        if view.is_cuda and _host_staged():
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            view.copy_(host, non_blocking=True)
        elif asynchronous:
            # RCCL's stream waits for the current (weight-gradient) stream up to here; nothing waits for RCCL until
            # sync_gradients calls wait() on the main stream
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)

    def _fire_ready(self):
        while self.fired < len(self.buckets) and self.buckets[self.fired].left == 0:
            b = self.buckets[self.fired]
            self._reduce_range(b.start, b.end, True)
            self.fired += 1
            self.stats["early"] += 1

    def on_grad_issued(self, w):
        b = self.bucket_of.get(id(w))
        if b is None:
            return
        n = self.seen.get(id(w), 0) + 1
        self.seen[id(w)] = n
        if not self.armed:
            return
        want = self.expected.get(id(w), 0)
        if n == want:
            b.left -= 1
            self._fire_ready()
        elif n > want and self.buckets.index(b) < self.fired:
            raise RuntimeError("FlatGrads: a weight received a gradient contribution after its bucket had been all-reduced "
                               "(the step's structure changed); call flatten_grads(module).relearn() when it does")

    def relearn(self):
        self.expected, self.stable, self.armed = None, 0, False

    def finish_step(self):
        """Everything not exchanged yet, in order; then the current stream waits for the buckets fired earlier."""
        while self.fired < len(self.buckets):
            b = self.buckets[self.fired]
            self._reduce_range(b.start, b.end, False)
            self.fired += 1
            self.stats["late"] += 1
        if self.tail_start < self.flat.numel():
            self._reduce_range(self.tail_start, self.flat.numel(), False)
        for w in self.works:
            w.wait()
        self.works = []
        self.stats["steps"] += 1
        if self.buckets:
            if self.expected is not None and self.expected == self.seen and self.seen:
                self.stable += 1
            else:
                self.expected, self.stable = dict(self.seen), 0
            self.armed = self.stable >= 1


_ACTIVE = None          # the FlatGrads whose backward pass is in progress (between zero_gradients and sync_gradients)


def _on_grad_issued(w):
    if _ACTIVE is not None:
        _ACTIVE.on_grad_issued(w)


def flatten_grads(module, bucket_bytes=0):
    """Give `module` a persistent flat gradient buffer (idempotent).  The train steps then keep `.grad` instead of
    resetting it to None (`zero_gradients`), and `sync_gradients` reduces the buffer in place; pair it with
    optim.Adam(..., grad_scale=1 / world_size(), zero_grads=True).  `bucket_bytes` > 0: exchange the leading buckets of
    that size from inside the backward pass (see FlatGrads)."""
    fg = getattr(module, '_sempyr_flat_grads', None)
    if fg is None:
        fg = FlatGrads(module, bucket_bytes)
        module._sempyr_flat_grads = fg
        if fg.buckets:
            from . import ops
            ops._GRAD_ISSUED_HOOK = _on_grad_issued
    return fg


def zero_gradients(module):
    """`module.zero_grad()` of the train steps (train_model.py:115,141).  With flat gradients the buffer already holds
    zeros at this point (written by the optimizer's last step, or by flatten_grads) and stays in place."""
    global _ACTIVE
    fg = getattr(module, '_sempyr_flat_grads', None)
    if fg is None:
        module.zero_grad()
    else:
        fg.reattach()
        if fg.buckets and _REDUCER is None and is_distributed():
            fg.begin_step()
            _ACTIVE = fg


_REDUCER = None          # test hook: callable(flat tensor) standing for the all-reduce (sum) of a 1-process rehearsal


def sync_gradients(module):
    """Sum (flat gradients: the optimizer divides) or average `module`'s parameter gradients over the data-parallel ranks;
    no-op for a single process.  Called between backward and optimizer.step() by the train steps."""
    global _ACTIVE
    fg = getattr(module, '_sempyr_flat_grads', None)
    if _REDUCER is not None and fg is not None:
        _REDUCER(fg.flat)
        return
    if not is_distributed():
        return
    if fg is not None and _ACTIVE is fg:
        _ACTIVE = None
        fg.finish_step()
        return
    if fg is not None:
        flat = fg.flat
        if flat.is_cuda and _host_staged():
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            # RCCL runs the collective on its own stream behind everything queued on this one and orders this stream after
            # it; the prefetch stream (next iteration's classifier taps + generator forward) keeps the GPU busy meanwhile
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return
    grads = []
    for p in module.parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)       # keep the collective's shape identical on all ranks
        grads.append(p.grad)
    allreduce_mean_(grads)
