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


class FlatGrads:
    """Persistent flat fp32 gradient buffer of a module; every `param.grad` is a view of it (see the module docstring)."""

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params or any(p.dtype != torch.float32 or not p.is_contiguous() for p in params):
            raise ValueError("flatten_grads: contiguous fp32 parameters expected")
        dev = params[0].device
        # every view starts on a 16-byte boundary (float4 accesses of the Adam / accumulate kernels)
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params, self.views = params, []
        for p, off in zip(params, offs):
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)

    def reattach(self):
        """put the views back after something replaced `.grad` (zero_grad(set_to_none=True), a foreign optimizer)"""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v


def flatten_grads(module):
    """Give `module` a persistent flat gradient buffer (idempotent).  The train steps then keep `.grad` instead of
    resetting it to None (`zero_gradients`), and `sync_gradients` reduces the buffer in place; pair it with
    optim.Adam(..., grad_scale=1 / world_size(), zero_grads=True)."""
    fg = getattr(module, '_sempyr_flat_grads', None)
    if fg is None:
        fg = FlatGrads(module)
        module._sempyr_flat_grads = fg
    return fg


def zero_gradients(module):
    """`module.zero_grad()` of the train steps (train_model.py:115,141).  With flat gradients the buffer already holds
    zeros at this point (written by the optimizer's last step, or by flatten_grads) and stays in place."""
    fg = getattr(module, '_sempyr_flat_grads', None)
    if fg is None:
        module.zero_grad()
    else:
        fg.reattach()


_REDUCER = None          # test hook: callable(flat tensor) standing for the all-reduce (sum) of a 1-process rehearsal


def sync_gradients(module):
    """Sum (flat gradients: the optimizer divides) or average `module`'s parameter gradients over the data-parallel ranks;
    no-op for a single process.  Called between backward and optimizer.step() by the train steps."""
    fg = getattr(module, '_sempyr_flat_grads', None)
    if _REDUCER is not None and fg is not None:
        _REDUCER(fg.flat)
        return
    if not is_distributed():
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
