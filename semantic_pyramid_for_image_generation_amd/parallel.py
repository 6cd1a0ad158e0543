"""Data parallelism: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on ROCm; "gloo"
for the CPU tests), identical parameters on every rank and ONE all-reduce of a flat fp32 gradient
buffer per optimizer step.  The reference has no distributed code at all; this is
the only collective on the path.  BatchNorm statistics and the diversity ratio stay per replica.

The flat buffer is a persistent tensor per module (7.0 M floats for D, 9.5 M for G): gradients are
packed with one multi-tensor copy, reduced with a single large collective (xGMI links are
point-to-point, so one 28-38 MB ring/direct exchange beats ~100 small ones), scaled by 1/N and
unpacked in place.
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
    torch.cuda.set_device(local)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        dist.init_process_group(backend or 'nccl', device_id=torch.device('cuda', local))
    return torch.device('cuda', local)


def broadcast_module(module, src=0):
    """Make parameters and buffers identical on all ranks (rank `src` wins)."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def _flat_buffer(params):
    key = tuple(id(p) for p in params)
    buf = _FLAT.get(key)
    n = sum(p.numel() for p in params)
    if buf is None or buf.numel() != n or buf.device != params[0].device:
        buf = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        _FLAT.clear() if len(_FLAT) > 8 else None
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
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / ws)
    torch._foreach_copy_(tensors, views)


def sync_gradients(module):
    """Average `module`'s parameter gradients over the data-parallel ranks (no-op for a single process).
    Called between backward and optimizer.step() by the train steps."""
    if not is_distributed():
        return
    grads = []
    for p in module.parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)       # keep the collective's shape identical on all ranks
        grads.append(p.grad)
    allreduce_mean_(grads)
