"""Differentiable operators of the hot path, each a thin `torch.autograd.Function` over the C ABI.

PyTorch supplies device memory (caching allocator), streams and the autograd *graph*; every
tensor computation is a hand-written HIP kernel in libsempyr_hip.so.  Each backward is itself
expressed with differentiable Functions, so the WGAN-GP double backward
(reference train_model.py:69-76, `create_graph=True`) works through every operator that the
discriminator uses.  There is no CPU path: non-CUDA(HIP) tensors raise.
"""
import ctypes
import os
import threading

import torch
from torch.autograd import Function

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, CONV_DGRAD, CONV_FWD, CONV_WGRAD, NORM_BATCH,
                   NORM_INSTANCE, NORM_LAYER, check)

L = _lib.lib

__all__ = ["conv2d", "conv_transpose2d", "norm_act", "norm_add_norm_act", "bn_eval_act", "act", "add_act", "mul", "normalize",
           "avg_pool2", "upsample2", "max_pool3s2", "global_avg_pool", "mean", "l1_mean", "gp_loss", "lerp_rows",
           "input_grads_only", "async_param_grads", "join_side_streams", "ACT_NONE", "ACT_RELU", "ACT_LRELU", "ACT_TANH", "NORM_BATCH", "NORM_INSTANCE",
           "NORM_LAYER"]


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _stream():
    """Raw handle of the current HIP stream of the current device.  (torch.cuda.current_stream() costs ~20 us of host
    time per call in this torch build — an os.environ lookup inside its device-index helper — which at ~4000 kernel
    launches per training cycle was a third of the host's issue time.)"""
    return _raw_stream(_cur_device())


class _NoCtx:
    """Stand-in for the autograd context when nothing is recorded."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _apply(fn, *args):
    """fn.apply(*args); with autograd off (no_grad forwards, and every operator call made from inside a plain backward
    pass) the Function's forward runs directly — no graph bookkeeping, ~10 us less host time per call."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoCtx(), *args)


def _req(t, name="tensor"):
    if not t.is_cuda:
        raise _lib.SempyrError(f"{name}: the sempyr operators run on the GPU only (got a {t.device} tensor); "
                               "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.SempyrError(f"{name}: fp32 expected, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return 0 if t is None else t.data_ptr()


_REDUCE_SCRATCH = L.sempyr_reduce_scratch_floats()

# HIP-graph capture (loop.py).  CAPTURE_EPOCH is 0 in eager execution and a unique positive id while a loop body is being
# captured.  Events recorded during a capture only exist inside that capture: cached objects that carry an event (packed
# weights) remember the epoch it was recorded in, and a reader only waits on the event when it runs in the same epoch --
# anything older is complete by construction (the device is synchronised before a capture begins, and graph launches are
# ordered on the launching stream).
CAPTURE_EPOCH = 0
_EPOCH_COUNTER = [0]


class capture_epoch:
    def __enter__(self):
        global CAPTURE_EPOCH
        _EPOCH_COUNTER[0] += 1
        CAPTURE_EPOCH = _EPOCH_COUNTER[0]
        return self

    def __exit__(self, *exc):
        global CAPTURE_EPOCH
        CAPTURE_EPOCH = 0

# When set, parameter (weight / bias / affine) gradients are skipped in backward passes. Used by
# gradient_penalty (only d pred / d x is wanted, train_model.py:73) — custom Functions cannot see
# autograd.grad's `inputs=` pruning, so the caller states it.  A plain module global: the autograd
# engine runs backward on its own device thread.
_INPUT_GRADS_ONLY = False


class input_grads_only:
    def __enter__(self):
        global _INPUT_GRADS_ONLY
        self._prev = _INPUT_GRADS_ONLY
        _INPUT_GRADS_ONLY = True

    def __exit__(self, *exc):
        global _INPUT_GRADS_ONLY
        _INPUT_GRADS_ONLY = self._prev


# Weight / bias gradients are leaves of the backward pass: nothing downstream waits for them until the optimizer
# step.  Inside `with async_param_grads():` they are computed on a second HIP stream and accumulated straight into
# `param.grad`, so the MFMA-bound WGRAD kernels overlap the HBM-bound normalisation / activation backward and the input-
# gradient chain on the main stream.  Leaving the block makes the current stream wait for the side stream.  Only plain
# backward passes (no create_graph) take this route; SEMPYR_ASYNC_WGRAD=0 turns it off.
_ASYNC_ENV = os.environ.get("SEMPYR_ASYNC_WGRAD", "1") != "0"
_ASYNC_ON = False
# measurement aid (bench.py's single-stream passes): the same kernels -- gradients accumulated by the combine kernels, no
# AccumulateGrad adds -- issued on the CURRENT stream, so that a kernel's event-to-event time is its own duration
SIDE_ON_MAIN = False
_SIDE = {}
_SIDE_DIRTY = set()


def _side_stream(device):
    s = _SIDE.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE[device.index] = s
    return s


def join_side_streams():
    for idx in list(_SIDE_DIRTY):
        torch.cuda.current_stream(idx).wait_stream(_SIDE[idx])
    _SIDE_DIRTY.clear()


class async_param_grads:
    """`with async_param_grads(): loss.backward()` -- inside the block the backward pass hands parameter gradients to this
    module instead of autograd's AccumulateGrad nodes: weight / bias gradients are computed on the side stream straight into
    `param.grad` (see above), and the small affine gradients of the norm layers are collected and written into `param.grad`
    by ONE launch when the block ends (autograd would issue an add kernel per tensor and pass)."""

    def __enter__(self):
        global _ASYNC_ON, _STASH_ON
        self._prev = (_ASYNC_ON, _STASH_ON)
        _ASYNC_ON = _ASYNC_ENV
        _STASH_ON = True
        return self

    def __exit__(self, *exc):
        global _ASYNC_ON, _STASH_ON
        _ASYNC_ON, _STASH_ON = self._prev
        join_side_streams()
        if not _STASH_ON:
            flush_param_grads()


_STASH_ON = False
_PENDING = []            # (param, gradient tensor) in the order the backward pass produced them


def _stash(param, g):
    """True when the gradient was taken over (the caller then returns None to autograd for that input)."""
    if not (_STASH_ON and g is not None and not torch.is_grad_enabled() and _is_leaf_param(param)):
        return False
    _PENDING.append((param, g, _current_stream_obj(g.device)))      # remember the stream the gradient is produced on
    return True


def flush_param_grads():
    """param.grad (+)= the collected gradients of every parameter, summed in production order: one launch per 32 parameters
    (up to four contributions each; a parameter with more goes into a later launch, so no two items of a launch share a
    destination) on the current stream, which first waits for every stream a contribution was produced on (no
    AccumulateGrad node runs for these gradients, so the autograd engine joins nothing)."""
    if not _PENDING:
        return
    pend = list(_PENDING)
    _PENDING.clear()
    dev = pend[0][1].device
    cur = _current_stream_obj(dev)
    for producer in {id(st): st for _, _, st in pend}.values():
        if producer != cur:             # no AccumulateGrad node runs for these gradients, so the engine joins nothing: do it here
            cur.wait_stream(producer)
    groups = {}                         # parameter -> its gradient contributions, in production order
    for p, g, producer in pend:
        g = _req(g, "parameter gradient")
        if g.numel() != p.numel():
            raise _lib.SempyrError("flush_param_grads: gradient / parameter size mismatch")
        if producer != cur:
            g.record_stream(cur)
        groups.setdefault(id(p), (p, []))[1].append(g)
    while groups:
        batch = list(groups.values())[:32]
        arr = (_lib.AccumItem * len(batch))()
        for i, (p, gs) in enumerate(batch):
            take, rest = gs[:4], gs[4:]
            dst = _grad_in_place(p)
            it = arr[i]
            if dst is None:
                if p.grad is not None:
                    raise _lib.SempyrError("flush_param_grads: .grad of a parameter is not a dense fp32 tensor of its shape")
                dst = torch.empty_like(p)
                p.grad = dst
                it.assign = 1
            else:
                it.assign = 0
            it.dst = dst.data_ptr(); it.numel = p.numel()
            for k in range(4):
                it.src[k] = take[k].data_ptr() if k < len(take) else None
            if rest:
                groups[id(p)] = (p, rest)        # more than four contributions: the rest goes into a later launch
            else:
                del groups[id(p)]
        check(L.sempyr_accumulate_many(arr, len(batch), _raw_stream(dev.index)), "sempyr_accumulate_many")


TWO_STREAM_CRITIC = os.environ.get("SEMPYR_TWO_STREAM", "1") != "0"
TWO_STREAM_GENERATOR = os.environ.get("SEMPYR_TWO_STREAM_G", "1") != "0"
PREFETCH = os.environ.get("SEMPYR_PREFETCH", "1") != "0"


def side_stream(device):
    """The library's second stream on `device` (independent branches of a step may run there, see train_discriminator)."""
    return _side_stream2(device)


_SIDE2 = {}


def _side_stream2(device):
    s = _SIDE2.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE2[device.index] = s
    return s


_PRE = {}


def prefetch_stream(device):
    """Third stream: work of the NEXT critic step that does not depend on the current one (see train_model.prefetch)."""
    s = _PRE.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _PRE[device.index] = s
    return s


def consumer_streams(device):
    """Every stream of this library on `device` that may read a tensor handed over from the prefetch stream."""
    return [torch.cuda.current_stream(device), _side_stream(device), _side_stream2(device)]


def _is_leaf_param(t):
    return t is not None and t.is_leaf and t.requires_grad


def _weak(t):
    """a callable returning `t` while it is alive (Function contexts keep parameters they only need for .grad this way)"""
    import weakref
    try:
        return weakref.ref(t)
    except TypeError:       # pragma: no cover
        return lambda: t


def _accumulate_grad(param, g, main):
    g.record_stream(main)                 # allocated on the side stream, consumed by the optimizer on the main one
    if param.grad is None:
        param.grad = g
    else:
        param.grad.add_(g)


_INPLACE_GRAD = os.environ.get("SEMPYR_INPLACE_GRAD", "1") != "0"


def _grad_in_place(param):
    """param.grad when a kernel may add to it in place (a plain dense fp32 tensor of the parameter's shape), else None."""
    g = param.grad
    if (_INPLACE_GRAD and g is not None and g.dtype == torch.float32 and g.is_cuda and g.is_contiguous() and g.shape == param.shape
            and not g.requires_grad):
        return g
    return None


_STREAM_OBJ = {}


def _current_stream_obj(device):
    """torch.cuda.current_stream(device) through a raw-handle cache (the torch call costs ~20 us, see _stream)."""
    raw = _raw_stream(device.index)
    st = _STREAM_OBJ.get((device.index, raw))
    if st is None:
        st = torch.cuda.current_stream(device)
        _STREAM_OBJ[(device.index, raw)] = st
    return st


# Data parallel (parallel.FlatGrads): called with the weight right after one of its gradient contributions has been issued
# on the side stream, still with that stream current -- a bucket of the flat gradient buffer whose last contribution this
# was is all-reduced from here, under the rest of the backward pass.
_GRAD_ISSUED_HOOK = None


def _param_grads_async(gy, x, plan, w, bias, want_w, want_b, bsrc=None):
    """WGRAD(gy, x) -> w.grad and channel_sum(gy) -> bias.grad on the side stream."""
    dev = gy.device
    main = _current_stream_obj(dev)
    side = main if SIDE_ON_MAIN else _side_stream(dev)
    if side is not main:
        side.wait_stream(main)
    torch.cuda.set_stream(side)            # (the `with torch.cuda.stream(...)` context costs ~40 us of host time)
    try:
        # a weight used by several passes of one step (D(real,fake), D(x_hat)): later contributions are added to .grad by
        # the reduction kernel itself instead of a separate add launch
        if want_w:
            cur = _grad_in_place(w)
            if cur is not None:
                _raw_conv_wgrad(plan, gy, x, into=cur)
            else:
                _accumulate_grad(w, _raw_conv_wgrad(plan, gy, x), main)
        if want_b:
            src = gy if bsrc is None else bsrc
            cur = _grad_in_place(bias)
            if cur is not None:
                _channel_sum_raw(src, into=cur)
            else:
                _accumulate_grad(bias, _channel_sum_raw(src), main)
            if bsrc is not None:
                bsrc.record_stream(side)
        if want_w and _GRAD_ISSUED_HOOK is not None:
            _GRAD_ISSUED_HOOK(w)
    finally:
        torch.cuda.set_stream(main)
    if side is not main:
        gy.record_stream(side)
        x.record_stream(side)
        _SIDE_DIRTY.add(dev.index)


def _can_async(w, bias, want_w, want_b):
    return (_ASYNC_ON and not torch.is_grad_enabled() and (want_w or want_b)
            and (not want_w or _is_leaf_param(w)) and (not want_b or _is_leaf_param(bias)))


# ------------------------------------------------------------------------------------------------
# convolution plans
# ------------------------------------------------------------------------------------------------
class _ModePlan:
    __slots__ = ("info", "ktab", "wsrc", "ws_bytes")


class ConvPlan:
    """Geometry + device tables of one Conv2d shape (cached per shape and device)."""

    def __init__(self, key, device, math=0):
        B, Cin, Hin, Win, Cout, kh, kw, stride, pad, ups = key
        self.key = key
        self.device = device
        self.math = math
        self.desc = _lib.ConvDesc(B, Cin, Hin, Win, Cout, kh, kw, stride, pad, ups, math)
        self.desc_ref = ctypes.byref(self.desc)
        self.modes = {}
        info = self.mode(CONV_FWD).info
        self.Ho, self.Wo = info.Ho, info.Wo
        self.B, self.Cin, self.Hin, self.Win, self.Cout = B, Cin, Hin, Win, Cout
        self.kh, self.kw, self.stride, self.pad, self.ups = kh, kw, stride, pad, ups
        self._noups = None

    def mode(self, mode):
        mp = self.modes.get(mode)
        if mp is None:
            mp = _ModePlan()
            mp.info = _lib.ConvPlanInfo()
            check(L.sempyr_conv2d_plan(self.desc_ref, mode, ctypes.byref(mp.info)), "sempyr_conv2d_plan")
            n = mp.info.tab_entries
            tab_mode = CONV_FWD if mode == CONV_WGRAD else mode
            kt = torch.empty(max(2 * n, 2), dtype=torch.int32)
            ws = torch.empty(max(n, 1), dtype=torch.int32)
            check(L.sempyr_conv2d_tables(self.desc_ref, tab_mode, kt.data_ptr(), ws.data_ptr()), "sempyr_conv2d_tables")
            mp.ktab = kt.to(self.device)
            mp.wsrc = ws.to(self.device)
            mp.ws_bytes = int(mp.info.workspace_bytes)
            self.modes[mode] = mp
        return mp

    def noups(self):
        """The same convolution seen on the already-upsampled input (for the input gradient)."""
        if not self.ups:
            return self
        if self._noups is None:
            B, Cin, Hin, Win, Cout, kh, kw, stride, pad, _ = self.key
            self._noups = get_plan((B, Cin, Hin * 2, Win * 2, Cout, kh, kw, stride, pad, 0), self.device)
        return self._noups


_PLANS = {}
_PLANS_LOCK = threading.Lock()
# Arithmetic of the convolutions (sempyr_conv_desc.math).  0: exact fp32 MFMA everywhere (the measured / default path).
# 1 (SEMPYR_CONV_MATH=bf16x3, opt-in): 3x3 stride-1 convolutions and their input gradients form every product from split-bf16
# operands on the bf16 matrix pipe (~2^-17 relative error per product); everything else stays fp32.  Read when a plan is
# looked up, so it can be switched at run time (tests, bench.py's separate bf16x3 figure).
CONV_MATH = _lib.MATH_BF16X3 if os.environ.get("SEMPYR_CONV_MATH", "f32").lower() in ("bf16x3", "1") else _lib.MATH_F32


def get_plan(key, device):
    k = (key, device.index, CONV_MATH)
    p = _PLANS.get(k)
    if p is None:
        with _PLANS_LOCK:
            p = _PLANS.get(k)
            if p is None:
                p = ConvPlan(key, device, CONV_MATH)
                _PLANS[k] = p
    return p


def _packed_weights(plan, w, mode):
    """k-major re-layout of a weight tensor; cached on leaf tensors (parameters) per version."""
    mp = plan.mode(mode)
    wkey = (mode, plan.key[1:])          # layout does not depend on the batch size
    cache = None
    if w.is_leaf:
        cache = getattr(w, "_sempyr_pack", None)
        if cache is None:
            cache = {}
            try:
                w._sempyr_pack = cache
            except Exception:  # pragma: no cover - tensors always accept attributes
                cache = None
        if cache is not None:
            hit = cache.get(wkey)
            if hit is not None and hit[0] == w._version and hit[2] == w.data_ptr():
                if hit[4] != _stream() and hit[7] == CAPTURE_EPOCH:   # packed on another stream: order this stream after the pack kernel
                    cur = torch.cuda.current_stream(w.device)
                    cur.wait_event(hit[3])
                    hit[1].record_stream(cur)
                return hit[1]
    wp = torch.empty(max(int(mp.info.wpack_floats), 1), dtype=torch.float32, device=w.device)
    check(L.sempyr_conv2d_pack_weights(plan.desc_ref, mode, w.data_ptr(), mp.wsrc.data_ptr(), wp.data_ptr(), _stream()),
          "sempyr_conv2d_pack_weights")
    if cache is not None:
        cur = torch.cuda.current_stream(w.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        cache[wkey] = (w._version, wp, w.data_ptr(), ev, cur.cuda_stream, plan, mode, CAPTURE_EPOCH)
    return wp


_REPACK = os.environ.get("SEMPYR_REPACK", "1") != "0"


def repack_cached(params, table_cache=None):
    """Refresh every cached packed layout of `params` (tensors an optimizer has just updated in place) with ONE launch on
    the current stream, into the buffers the cache already holds.  Without it the first convolution that touches a weight
    after the step re-packs it with a launch of its own (~100 small launches per cycle at the head of each pass).
    Must run stream-ordered after every reader of the old weights — which the in-place optimizer step requires anyway.
    `table_cache`: a dict the caller keeps, so the ctypes item table is rebuilt only when the set of layouts changes."""
    if not _REPACK:
        return
    entries = []
    sig = []
    for w in params:
        cache = getattr(w, "_sempyr_pack", None)
        if not cache:
            continue
        ptr = w.data_ptr()
        for wkey, hit in cache.items():
            if hit[2] == ptr:
                entries.append((w, wkey, hit))
                sig.append((ptr, hit[1].data_ptr(), hit[6]))
    if not entries:
        return
    dev = entries[0][0].device
    sig = tuple(sig)
    arr = None
    if table_cache is not None and table_cache.get("sig") == sig:
        arr = table_cache["arr"]
    if arr is None:
        arr = (_lib.PackItem * len(entries))()
        for i, (w, wkey, hit) in enumerate(entries):
            plan, mode = hit[5], hit[6]
            it = arr[i]
            it.desc = plan.desc
            it.mode = mode
            it.w = w.data_ptr(); it.wsrc = plan.mode(mode).wsrc.data_ptr(); it.wpacked = hit[1].data_ptr()
        if table_cache is not None:
            table_cache["sig"] = sig; table_cache["arr"] = arr
    raw = _raw_stream(dev.index)
    check(L.sempyr_conv2d_pack_weights_many(arr, len(entries), raw), "sempyr_conv2d_pack_weights_many")
    cur = _current_stream_obj(dev)
    ev = torch.cuda.Event()
    ev.record(cur)
    for w, wkey, hit in entries:
        w._sempyr_pack[wkey] = (w._version, hit[1], hit[2], ev, raw, hit[5], hit[6], CAPTURE_EPOCH)


def _workspace(nbytes, device):
    if nbytes <= 0:
        return None
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _raw_conv_fwd(plan, x, w, bias, act_kind=ACT_NONE, slope=0.0, pooled=None, scale_ptr=0, res=None, res_half=False,
                  bits=None, mask_bits=None, want_y=True):
    """`bias` is a tensor or a raw device address (the shift row of cached BatchNorm coefficients).  `bits` (out) /
    `mask_bits` (in): the sign bits of the fused pooled epilogue (see sempyr_conv2d_fwd); want_y=False: pooled output only."""
    mp = plan.mode(CONV_FWD)
    wp = _packed_weights(plan, w, CONV_FWD)
    y = torch.empty((plan.B, plan.Cout, plan.Ho, plan.Wo), dtype=torch.float32, device=x.device) if want_y else None
    ws = _workspace(mp.ws_bytes, x.device)
    check(L.sempyr_conv2d_fwd(plan.desc_ref, x.data_ptr(), wp.data_ptr(), mp.ktab.data_ptr(),
                              bias if isinstance(bias, int) else _p(bias), scale_ptr, _p(res), 1 if res_half else 0, act_kind, slope,
                              _p(y), _p(pooled), _p(bits), _p(mask_bits), _p(ws), mp.ws_bytes, _stream()), "sempyr_conv2d_fwd")
    return y


_CONV_STATS = os.environ.get("SEMPYR_CONV_STATS", "1") != "0"
_PENDING_STATS = []


def conv_stats_granule(plan):
    """Pixels per group of the BatchNorm-statistics epilogue of `plan`'s forward convolution; 0 = it has none."""
    return int(plan.mode(CONV_FWD).info.stats_granule) if _CONV_STATS else 0


def _raw_conv_fwd_stats(plan, x, w, granule):
    """y = conv(x, w) and the per-group (sum, M2) partials of y for the BatchNorm that follows (sempyr_conv2d_fwd_stats)."""
    mp = plan.mode(CONV_FWD)
    wp = _packed_weights(plan, w, CONV_FWD)
    y = torch.empty((plan.B, plan.Cout, plan.Ho, plan.Wo), dtype=torch.float32, device=x.device)
    ts = torch.empty(2 * plan.B * plan.Cout * (plan.Ho * plan.Wo // granule), dtype=torch.float32, device=x.device)
    check(L.sempyr_conv2d_fwd_stats(plan.desc_ref, x.data_ptr(), wp.data_ptr(), mp.ktab.data_ptr(), y.data_ptr(), ts.data_ptr(),
                                    _stream()), "sempyr_conv2d_fwd_stats")
    return y, ts


def tile_stats_of(t):
    """(partials, granule) a convolution's epilogue left for the BatchNorm statistics of `t`, or None."""
    return getattr(t, "_sempyr_tile_stats", None)


_STEM_BITS = os.environ.get("SEMPYR_STEM_BITS", "1") != "0"


def _pooled_bits_ok(plan, act_kind):
    """True when conv -> ReLU-family activation -> AvgPool2d(2) can keep sign bits instead of the full-resolution output."""
    return (_STEM_BITS and act_kind in (ACT_RELU, ACT_LRELU) and plan.Cout > 4 and (plan.Ho * plan.Wo) % 32 == 0
            and plan.Ho % 2 == 0 and plan.Wo % 4 == 0 and plan.mode(CONV_FWD).info.splits[0] <= 1)


_STEM_GRADS = os.environ.get("SEMPYR_STEM_GRADS", "1") != "0"


def _stem_grads_ok(plan):
    """The critic stem at a width its dedicated gradient kernels are built for (discriminator.py:105 on 256-pixel rows):
    weight, bias and input gradients then come straight from (pooled gradient, sign bits)."""
    return (_STEM_GRADS and plan.Cin == 3 and plan.Cout == 64 and plan.kh == 3 and plan.kw == 3 and plan.stride == 1
            and plan.pad == 1 and not plan.ups and plan.Wo == 256 and plan.Ho % 2 == 0 and plan.mode(CONV_DGRAD).info.splits[0] <= 1)


def _raw_stem_dgrad(plan, g, bits, slope, w):
    """conv_transpose(0.25 * up2(g) * act'(bits), w) without the full-resolution intermediate (sempyr_conv2d_dgrad_unpooled)."""
    mp = plan.mode(CONV_DGRAD)
    wp = _packed_weights(plan, w, CONV_DGRAD)
    gx = torch.empty((plan.B, plan.Cin, plan.Hin, plan.Win), dtype=torch.float32, device=g.device)
    check(L.sempyr_conv2d_dgrad_unpooled(plan.desc_ref, g.data_ptr(), bits.data_ptr(), slope, wp.data_ptr(), mp.ktab.data_ptr(),
                                         gx.data_ptr(), _stream()), "sempyr_conv2d_dgrad_unpooled")
    return gx


def _raw_stem_wgrad(plan, g, bits, slope, x, gw_into=None, want_b=False, gb_into=None):
    """(gw, gb) of the stem from (pooled gradient, sign bits, input); `*_into`: existing gradients added to in place."""
    nbytes = getattr(plan, "_stem_ws", None)
    if nbytes is None:
        out = ctypes.c_size_t(0)
        check(L.sempyr_stem_wgrad_workspace(plan.desc_ref, ctypes.addressof(out)), "sempyr_stem_wgrad_workspace")
        nbytes = plan._stem_ws = int(out.value)
    ws = _workspace(nbytes, g.device)
    gw = gw_into if gw_into is not None else torch.empty((plan.Cout, plan.Cin, plan.kh, plan.kw), dtype=torch.float32, device=g.device)
    gb = None
    if want_b:
        gb = gb_into if gb_into is not None else torch.empty(plan.Cout, dtype=torch.float32, device=g.device)
    check(L.sempyr_stem_wgrad(plan.desc_ref, g.data_ptr(), bits.data_ptr(), slope, x.data_ptr(), gw.data_ptr(),
                              0 if gw_into is None else 1, _p(gb), 0 if gb_into is None else 1, _p(ws), nbytes, _stream()),
          "sempyr_stem_wgrad")
    return gw, gb


def _stem_param_grads(g, bits, slope, x, plan, w, bias, want_b):
    """Weight (+ bias) gradient of the stem; on the side stream straight into .grad when that is possible (returns None, None),
    else returns the tensors."""
    if not _can_async(w, bias, True, want_b):
        return _raw_stem_wgrad(plan, g, bits, slope, x, want_b=want_b)
    dev = g.device
    main = _current_stream_obj(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    torch.cuda.set_stream(side)
    try:
        cur_w = _grad_in_place(w)
        cur_b = _grad_in_place(bias) if want_b else None
        gw, gb = _raw_stem_wgrad(plan, g, bits, slope, x, cur_w, want_b, cur_b)
        if cur_w is None:
            _accumulate_grad(w, gw, main)
        if want_b and cur_b is None:
            _accumulate_grad(bias, gb, main)
        if _GRAD_ISSUED_HOOK is not None:
            _GRAD_ISSUED_HOOK(w)
    finally:
        torch.cuda.set_stream(main)
    g.record_stream(side)
    x.record_stream(side)
    bits.record_stream(side)
    _SIDE_DIRTY.add(dev.index)
    return None, None


def _raw_conv_dgrad(plan, gy, w, bias):
    """plan.ups must be 0 here (callers pass plan.noups())."""
    mp = plan.mode(CONV_DGRAD)
    wp = _packed_weights(plan, w, CONV_DGRAD)
    gx = torch.empty((plan.B, plan.Cin, plan.Hin, plan.Win), dtype=torch.float32, device=gy.device)
    ws = _workspace(mp.ws_bytes, gy.device)
    check(L.sempyr_conv2d_dgrad(plan.desc_ref, gy.data_ptr(), wp.data_ptr(), mp.ktab.data_ptr(), _p(bias), gx.data_ptr(),
                                _p(ws), mp.ws_bytes, _stream()), "sempyr_conv2d_dgrad")
    return gx


def _raw_conv_wgrad(plan, gy, x, into=None):
    """`into`: an existing gradient tensor the result is added to in place (by the combine pass, no extra launch)."""
    mp = plan.mode(CONV_WGRAD)
    gw = into if into is not None else torch.empty((plan.Cout, plan.Cin, plan.kh, plan.kw), dtype=torch.float32,
                                                   device=gy.device)
    ws = _workspace(mp.ws_bytes, gy.device)
    check(L.sempyr_conv2d_wgrad(plan.desc_ref, gy.data_ptr(), x.data_ptr(), mp.ktab.data_ptr(), gw.data_ptr(),
                                0 if into is None else 1, _p(ws), mp.ws_bytes, _stream()), "sempyr_conv2d_wgrad")
    return gw


def _channel_sum_raw(g, scale=1.0, into=None):
    N, C = g.shape[0], g.shape[1]
    HW = g.numel() // (N * C)
    out = into if into is not None else torch.empty(C, dtype=torch.float32, device=g.device)
    scratch = torch.empty(N * C, dtype=torch.float32, device=g.device)
    check(L.sempyr_channel_sum(g.data_ptr(), N, C, HW, scale, out.data_ptr(), 0 if into is None else 1, scratch.data_ptr(),
                               _stream()), "sempyr_channel_sum")
    return out


def _bias_grad(g):
    if torch.is_grad_enabled() and g.requires_grad:
        raise NotImplementedError("double backward through a bias gradient is not part of the hot path")
    return _channel_sum_raw(g)


class _ConvFwd(Function):
    """y = act(conv2d(x[, upsampled x2], w) + bias), the activation running in the kernel epilogue; with `pool`
    the Function returns AvgPool2d(2)(y) and its backward folds the pool's and the activation's backward into one pass
    (critic stem, discriminator.py:106-109)."""

    @staticmethod
    def forward(ctx, x, w, bias, plan, act_kind=ACT_NONE, slope=0.0, pool=False, stats=0):
        # graph_only_input_grad(x): the gradient with respect to x is wanted inside a create_graph pass only (the gradient
        # penalty's interpolated batch); a plain backward pass through this convolution then skips it
        ctx.dx_graph_only = getattr(x, '_sempyr_dx_graph_only', False)
        x = _req(x, "conv input"); w = _req(w, "conv weight")
        if bias is not None:
            bias = _req(bias, "conv bias")
        if act_kind == ACT_TANH and pool:
            raise _lib.SempyrError("conv2d: pooled output needs a ReLU-family (or no) activation")
        if stats:       # bias-free, no activation, no pool (checked by conv2d): the epilogue also leaves BatchNorm partials
            y, ts = _raw_conv_fwd_stats(plan, x, w, stats)
            _PENDING_STATS.append((ts, stats))
            ctx.save_for_backward(x, w, None, None)
            ctx.plan = plan
            ctx.has_bias = False
            ctx.act = (ACT_NONE, 0.0)
            ctx.pool = False
            ctx.bits = False
            return y
        out = bits = None
        if pool:
            out = torch.empty((plan.B, plan.Cout, plan.Ho // 2, plan.Wo // 2), dtype=torch.float32, device=x.device)
            if _pooled_bits_ok(plan, act_kind):
                # the backward only needs the SIGN of the full-resolution activation: 1 bit instead of 4 bytes per element
                bits = torch.empty(plan.B * plan.Cout * plan.Ho * plan.Wo // 32, dtype=torch.int32, device=x.device)
        y = _raw_conv_fwd(plan, x, w, bias, act_kind, slope, out, bits=bits, want_y=bits is None)
        ctx.save_for_backward(x, w, bits if bits is not None else (y if act_kind != ACT_NONE else None), bias)
        ctx.plan = plan
        ctx.has_bias = bias is not None
        ctx.act = (act_kind, slope)
        ctx.pool = pool
        ctx.bits = bits is not None
        return out if pool else y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, bias = ctx.saved_tensors
        plan = ctx.plan
        gy = gy.contiguous()
        want_w = ctx.needs_input_grad[1] and not _INPUT_GRADS_ONLY
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not _INPUT_GRADS_ONLY
        want_x = ctx.needs_input_grad[0] and not (ctx.dx_graph_only and not torch.is_grad_enabled())
        bsrc = None               # what the bias gradient is the channel sum of (default: gy itself)
        if ctx.pool:
            h, wd = gy.shape[2], gy.shape[3]
            yref, bits = (None, y) if ctx.bits else (y, None)
            if ctx.bits and want_x and not (want_w or want_b):
                # input gradient only (gradient penalty's inner pass, generator step): one Function whose own backward
                # (the penalty's double backward) is a single fused launch
                return _apply(_StemInputGrad, gy, w, bits, plan, *ctx.act), None, None, None, None, None, None, None
            if ctx.bits and want_w and not torch.is_grad_enabled() and _stem_grads_ok(plan) and gy.data_ptr() % 16 == 0:
                # loss.backward() through the stem (train_model.py:139): every gradient straight from the pooled gradient
                # and the sign bits, nothing full-resolution is materialised
                slope = ctx.act[1] if ctx.act[0] == ACT_LRELU else 0.0
                gy = _req(gy, "grad")
                gw, gb = _stem_param_grads(gy, bits, slope, x, plan, w, bias, want_b)
                gx = _raw_stem_dgrad(plan, gy, bits, slope, w) if want_x else None
                return gx, gw, gb, None, None, None, None, None
            if want_b and not torch.is_grad_enabled() and (h * wd) % 512 == 0 and wd % 2 == 0 and y.data_ptr() % 16 == 0:
                # the un-pool pass also emits per-row partial sums: the bias gradient needs no second pass over the
                # full-resolution gradient
                bsrc = torch.empty((gy.shape[0], gy.shape[1], (h * wd) // 512), dtype=torch.float32, device=gy.device)
                gy = _pool2_up_raw(_req(gy, "grad"), 0.25, yref, *ctx.act, partials=bsrc, bits=bits)
            else:
                gy = _apply(_Pool2Up, gy, 0.25, yref, ctx.act[0], ctx.act[1], bits)
        elif y is not None:
            gy = _apply(_ActBwd, gy, y, *ctx.act)
        gx = gw = gb = None
        if want_w or want_b:
            if _can_async(w, bias, want_w, want_b):
                _param_grads_async(gy, x, plan, w, bias, want_w, want_b, bsrc)     # issued first: overlaps the input gradient
            else:
                if want_w:
                    gw = _apply(_ConvWgrad, gy, x, plan)
                if want_b:
                    gb = _bias_grad(gy if bsrc is None else bsrc)
        if want_x:
            gx = _conv_input_grad(gy, w, plan)
        return gx, gw, gb, None, None, None, None, None


class _StemInputGrad(Function):
    """gx = conv_transpose(0.25 * up2(g) * act'(bits), w): the input gradient of `conv -> ReLU-family act -> AvgPool2d(2)` given the
    gradient g of the POOLED output and the sign bits of the activation.  Its backward (the gradient penalty's double
    backward through the critic stem) is one launch: conv(u, w) masked by the bits and pooled in the epilogue -- the
    full-resolution intermediate is never written."""

    @staticmethod
    def forward(ctx, g, w, bits, plan, act_kind, slope):
        g = _req(g, "grad"); w = _req(w, "conv weight")
        ctx.direct = _stem_grads_ok(plan) and g.data_ptr() % 16 == 0
        if ctx.direct:
            gx = _raw_stem_dgrad(plan, g, bits, slope if act_kind == ACT_LRELU else 0.0, w)
            ctx.save_for_backward(g, w, bits)
        else:
            gz = _pool2_up_raw(g, 0.25, None, act_kind, slope, bits=bits)
            gx = _raw_conv_dgrad(plan.noups(), gz, w, None)
            ctx.save_for_backward(gz, w, bits)
        ctx.plan = plan
        ctx.act = (act_kind, slope)
        return gx

    @staticmethod
    def backward(ctx, u):
        gz, w, bits = ctx.saved_tensors
        plan = ctx.plan
        act_kind, slope = ctx.act
        if torch.is_grad_enabled() and u.requires_grad:
            raise NotImplementedError("third-order derivatives through the critic stem are not part of the hot path")
        u = _req(u, "double-backward grad")
        g_g = g_w = None
        if not _INPUT_GRADS_ONLY and ctx.needs_input_grad[1]:
            if ctx.direct:          # `gz` is the pooled gradient here
                g_w, _ = _stem_param_grads(gz, bits, slope if act_kind == ACT_LRELU else 0.0, u, plan, w, None, False)
            elif _can_async(w, None, True, False):
                _param_grads_async(gz, u, plan, w, None, True, False)
            else:
                g_w = _raw_conv_wgrad(plan, gz, u)
        if ctx.needs_input_grad[0]:
            g_g = torch.empty((plan.B, plan.Cout, plan.Ho // 2, plan.Wo // 2), dtype=torch.float32, device=u.device)
            _raw_conv_fwd(plan, u, w, None, ACT_NONE, slope if act_kind == ACT_LRELU else 0.0, g_g, mask_bits=bits, want_y=False)
        return g_g, g_w, None, None, None, None


def _conv_input_grad(gy, w, plan):
    gx = _apply(_ConvDgrad, gy, w, None, plan.noups())
    if plan.ups:
        gx = _apply(_Pool2Down, gx, 1.0)
    return gx


class _ConvDgrad(Function):
    """gx = conv_transpose2d(gy, w) (+ bias): the input gradient of `plan`'s conv, and the forward of a
    ConvTranspose2d.  plan.ups == 0."""

    @staticmethod
    def forward(ctx, gy, w, bias, plan):
        gy = _req(gy, "dgrad input"); w = _req(w, "conv weight")
        if bias is not None:
            bias = _req(bias, "bias")
        ctx.save_for_backward(gy, w, bias)
        ctx.plan = plan
        ctx.has_bias = bias is not None
        return _raw_conv_dgrad(plan, gy, w, bias)

    @staticmethod
    def backward(ctx, u):
        gy, w, bias = ctx.saved_tensors
        plan = ctx.plan
        u = u.contiguous()
        g_gy = g_w = g_b = None
        if not _INPUT_GRADS_ONLY:
            want_w = ctx.needs_input_grad[1]
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if _can_async(w, bias, want_w, False) and not want_b:
                _param_grads_async(gy, u, plan, w, None, want_w, False)
            else:
                if want_w:
                    g_w = _apply(_ConvWgrad, gy, u, plan)
                if want_b:
                    g_b = _bias_grad(u)
        if ctx.needs_input_grad[0]:
            g_gy = _apply(_ConvFwd, u, w, None, plan)
        return g_gy, g_w, g_b, None


class _ConvWgrad(Function):
    """gw = sum_{b,pix} gy (x) x  — weight gradient of `plan`'s conv (x may be pre-upsample)."""

    @staticmethod
    def forward(ctx, gy, x, plan):
        gy = _req(gy, "wgrad gy"); x = _req(x, "wgrad x")
        ctx.save_for_backward(gy, x)
        ctx.plan = plan
        return _raw_conv_wgrad(plan, gy, x)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        plan = ctx.plan
        ggw = ggw.contiguous()
        g_gy = g_x = None
        if ctx.needs_input_grad[0]:
            g_gy = _apply(_ConvFwd, x, ggw, None, plan)
        if ctx.needs_input_grad[1]:
            g_x = _conv_input_grad(gy, ggw, plan)
        return g_gy, g_x, None


def graph_only_input_grad(x):
    """Mark a leaf that requires grad as differentiated inside `torch.autograd.grad(..., create_graph=True)` ONLY: a plain
    backward pass then does not compute the gradient with respect to it.  The WGAN-GP penalty (train_model.py:69-76) makes
    its interpolated batch such a leaf; the reference's `loss.backward()` goes on to compute d loss / d x-hat through the
    critic's first convolution and throws it away.  Parameter gradients do not depend on it."""
    x._sempyr_dx_graph_only = True
    return x


def conv2d(x, w, bias=None, stride=1, padding=0, upsample=False, act_kind=ACT_NONE, slope=0.0, pool=False, stats=False):
    """nn.Conv2d forward (reference generator.py:25, discriminator.py:37), optionally reading its input
    through a fused nn.UpsamplingNearest2d(2) (generator.py:44,78) and applying a following activation
    (discriminator.py:106-107, generator.py:116-120) in the kernel epilogue.  `pool`: also apply the
    nn.AvgPool2d(2) that follows (discriminator.py:109).  `stats`: a training-mode BatchNorm2d follows (generator.py:23-27);
    where the kernel has the epilogue for it, the result carries the statistics partials (tile_stats_of) and the norm skips
    its pass over the tensor."""
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = w.shape
    if Cin_w != Cin:
        raise _lib.SempyrError(f"conv2d: weight expects {Cin_w} input channels, got {Cin}")
    plan = get_plan((B, Cin, H, W, Cout, kh, kw, stride, padding, 1 if upsample else 0), x.device)
    if pool and (plan.Ho % 2 or plan.Wo % 2):
        raise _lib.SempyrError("conv2d: pooled output needs even output height and width")
    granule = conv_stats_granule(plan) if (stats and bias is None and act_kind == ACT_NONE and not pool) else 0
    if not granule:
        return _apply(_ConvFwd, x, w, bias, plan, act_kind, slope, pool)
    del _PENDING_STATS[:]
    y = _apply(_ConvFwd, x, w, bias, plan, act_kind, slope, pool, granule)
    y._sempyr_tile_stats = _PENDING_STATS.pop()
    return y


class _ZeroConv(Function):
    """The output of a bias-free convolution whose input is known to be all zeros: zeros, with a zero weight gradient.
    (The reference multiplies the classifier taps by masks that are all-zero for every level at or above the selected
    layer -- quirk Q2 -- and convolves the zeros; train_model.py:197-209, generator.py:126-129.)"""

    @staticmethod
    def forward(ctx, w, shape):
        w = _req(w, "conv weight")
        y = torch.empty(shape, dtype=torch.float32, device=w.device)
        check(L.sempyr_fill_const(y.data_ptr(), 0.0, y.numel(), _stream()), "sempyr_fill_const")
        ctx.wshape = tuple(w.shape)
        ctx.w_ref = _weak(w)
        return y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.needs_input_grad[0] or _INPUT_GRADS_ONLY:
            return None, None
        w = ctx.w_ref()
        gw = torch.empty(ctx.wshape, dtype=torch.float32, device=gy.device)
        check(L.sempyr_fill_const(gw.data_ptr(), 0.0, gw.numel(), _stream()), "sempyr_fill_const")
        if w is not None and _stash(w, gw):
            return None, None
        return gw, None


def bn_zero_input_add(a, beta, act_kind=ACT_NONE, slope=0.0, bn_state=None):
    """a + act(BatchNorm2d(0)) for a training-mode BatchNorm whose input is all zeros (sempyr_bn_zero_input_add): the
    batch statistics are exactly 0, the output is a + act(beta[c]), the running buffers decay.  No autograd."""
    if torch.is_grad_enabled():
        raise _lib.SempyrError("bn_zero_input_add is a forward-only helper: call it under torch.no_grad()")
    a = _req(a, "input"); beta = _req(beta, "norm bias")
    N, C, HW = _nchw(a)
    rm = rv = nbt = None
    momentum = 0.0
    if bn_state is not None:
        rm, rv, nbt, momentum = bn_state
    y = torch.empty_like(a)
    check(L.sempyr_bn_zero_input_add(a.data_ptr(), beta.data_ptr(), y.data_ptr(), N, C, HW, act_kind, slope, _p(rm), _p(rv), _p(nbt),
                                     momentum, _stream()), "sempyr_bn_zero_input_add")
    if bn_state is not None:
        torch.autograd.graph.increment_version(rm)
        torch.autograd.graph.increment_version(rv)
    return y


def zero_conv_output(w, shape):
    """conv(zeros, w) for a bias-free convolution: see _ZeroConv."""
    return _apply(_ZeroConv, w, tuple(shape))


def conv_transpose2d(x, w, bias=None, stride=1, padding=0):
    """nn.ConvTranspose2d forward (reference generator.py:23,115): the input gradient of the Conv2d
    (Cout_t -> Cin_t) that shares the weight tensor (Cin_t, Cout_t, kh, kw)."""
    B, Cin_t, H, W = x.shape
    Cin_w, Cout_t, kh, kw = w.shape
    if Cin_w != Cin_t:
        raise _lib.SempyrError(f"conv_transpose2d: weight expects {Cin_w} input channels, got {Cin_t}")
    Ho = (H - 1) * stride - 2 * padding + kh
    Wo = (W - 1) * stride - 2 * padding + kw
    plan = get_plan((B, Cout_t, Ho, Wo, Cin_t, kh, kw, stride, padding, 0), x.device)
    if plan.Ho != H or plan.Wo != W:
        raise _lib.SempyrError("conv_transpose2d: inconsistent geometry")
    return _apply(_ConvDgrad, x, w, bias, plan)


def linear(x, w, bias=None):
    """nn.Linear (ResNet18 fc, classifier.py:43) as a 1x1 convolution on a 1x1 image."""
    B, F = x.shape
    y = conv2d(x.reshape(B, F, 1, 1), w.reshape(w.shape[0], F, 1, 1), bias)
    return y.reshape(B, w.shape[0])


# ------------------------------------------------------------------------------------------------
# normalisation (+ fused activation)
# ------------------------------------------------------------------------------------------------
def _nchw(x):
    N, C = x.shape[0], x.shape[1]
    return N, C, x.numel() // (N * C)


def _norm_stats(x, gamma, beta, kind, eps, bn_state, rep=1, tiles=None):
    """per-row (mean, rstd, scale, shift) of a training-mode norm; updates the BatchNorm running statistics.
    `tiles` = (partials, granule) from the producing convolution's epilogue (BATCH only): no pass over x."""
    N, C, HW = _nchw(x)
    rows = N * C
    rowbuf = torch.empty(4 * rows, dtype=torch.float32, device=x.device)
    rm = rv = nbt = None
    momentum = 0.0
    if bn_state is not None:
        rm, rv, nbt, momentum = bn_state
    if tiles is not None and kind == NORM_BATCH:
        ts, granule = tiles
        if ts.numel() != 2 * rows * (HW // granule):
            raise _lib.SempyrError("norm: the statistics partials do not belong to this tensor")
        check(L.sempyr_bn_stats_from_tiles(ts.data_ptr(), N, C, HW, granule, eps, gamma.data_ptr(), beta.data_ptr(),
                                           _p(rm), _p(rv), _p(nbt), momentum, rep, rowbuf.data_ptr(), _stream()),
              "sempyr_bn_stats_from_tiles")
    else:
        scratch = torch.empty(2 * rows, dtype=torch.float32, device=x.device)
        check(L.sempyr_norm_stats(kind, x.data_ptr(), N, C, HW, eps, gamma.data_ptr(), beta.data_ptr(),
                                  _p(rm), _p(rv), _p(nbt), momentum, rep, rowbuf.data_ptr(), scratch.data_ptr(), _stream()),
              "sempyr_norm_stats")
    if bn_state is not None:      # the kernel updated the buffers through raw pointers: tell version-keyed caches
        torch.autograd.graph.increment_version(rm)
        torch.autograd.graph.increment_version(rv)
    return rowbuf


class _NormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, kind, act_kind, slope, eps, bn_state, tiles=None):
        x = _req(x, "norm input"); gamma = _req(gamma, "norm weight"); beta = _req(beta, "norm bias")
        N, C, HW = _nchw(x)
        rows = N * C
        y = torch.empty_like(x)
        if tiles is not None and kind == NORM_BATCH:
            # statistics from the producing convolution's epilogue, then the apply pass
            rowbuf = _norm_stats(x, gamma, beta, kind, eps, bn_state, 1, tiles)
            check(L.sempyr_affine_act(x.data_ptr(), rowbuf.data_ptr(), rowbuf.data_ptr() + 8 * rows, rowbuf.data_ptr() + 12 * rows, 0,
                                      y.data_ptr(), rows, HW, act_kind, slope, _stream()), "sempyr_affine_act")
        else:
            rowbuf = torch.empty(4 * rows, dtype=torch.float32, device=x.device)
            scratch = torch.empty(2 * rows, dtype=torch.float32, device=x.device)
            rm = rv = nbt = None
            momentum = 0.0
            if bn_state is not None:
                rm, rv, nbt, momentum = bn_state
            # statistics + apply in one call (one launch for InstanceNorm rows up to 16384 elements)
            check(L.sempyr_norm_fwd(kind, x.data_ptr(), N, C, HW, eps, gamma.data_ptr(), beta.data_ptr(), _p(rm), _p(rv), _p(nbt),
                                    momentum, act_kind, slope, y.data_ptr(), rowbuf.data_ptr(), scratch.data_ptr(), _stream()),
                  "sempyr_norm_fwd")
            if bn_state is not None:      # the kernel updated the buffers through raw pointers: tell version-keyed caches
                torch.autograd.graph.increment_version(rm)
                torch.autograd.graph.increment_version(rv)
        ctx.save_for_backward(x, y if act_kind != ACT_NONE else None, gamma, rowbuf)
        ctx.cfg = (kind, act_kind, slope)
        ctx.beta_ref = _weak(beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, rowbuf = ctx.saved_tensors
        kind, act_kind, slope = ctx.cfg
        gx, gg, gb = _apply(_NormActBwd, gy.contiguous(), x, y, gamma, rowbuf, kind, act_kind, slope)
        if _INPUT_GRADS_ONLY:
            gg = gb = None
        else:
            if _stash(gamma, gg):
                gg = None
            if _stash(ctx.beta_ref(), gb):
                gb = None
        return gx, gg, gb, None, None, None, None, None, None


class _NormAddNormAct(Function):
    """y = act(norm_a(a) + norm_b(b)): the tail of a residual block (generator.py:54-57) in one apply pass instead of
    two norm passes and an add.  The backward is composed of the single-norm pieces."""

    @staticmethod
    def forward(ctx, a, gamma_a, beta_a, state_a, b, gamma_b, beta_b, state_b, kind, act_kind, slope, eps_a, eps_b,
                b_half=False, tiles_a=None):
        a = _req(a, "norm input"); b = _req(b, "norm input")
        if b_half:
            # b is the branch BEFORE a nearest x2 upsample (its 1x1 convolution commutes with the upsample): statistics on
            # the low-resolution map (identical mean / biased variance; the unbiased running variance counts 4x the elements)
            if a.dim() != 4 or tuple(a.shape) != (b.shape[0], b.shape[1], 2 * b.shape[2], 2 * b.shape[3]):
                raise _lib.SempyrError("norm_add_norm_act: the half-resolution branch must have half the height and width")
        elif a.shape != b.shape:
            raise _lib.SempyrError("norm_add_norm_act: shape mismatch")
        gamma_a = _req(gamma_a, "norm weight"); beta_a = _req(beta_a, "norm bias")
        gamma_b = _req(gamma_b, "norm weight"); beta_b = _req(beta_b, "norm bias")
        N, C, HW = _nchw(a)
        rows = N * C
        ra = _norm_stats(a, gamma_a, beta_a, kind, eps_a, state_a, 1, tiles_a)
        rb = _norm_stats(b, gamma_b, beta_b, kind, eps_b, state_b, 4 if b_half else 1)
        y = torch.empty_like(a)
        check(L.sempyr_affine2_act(a.data_ptr(), ra.data_ptr(), ra.data_ptr() + 8 * rows, ra.data_ptr() + 12 * rows,
                                   b.data_ptr(), rb.data_ptr(), rb.data_ptr() + 8 * rows, rb.data_ptr() + 12 * rows,
                                   y.data_ptr(), rows, HW, act_kind, slope, a.shape[3] if b_half else 0, _stream()),
              "sempyr_affine2_act")
        ctx.save_for_backward(a, b, y if act_kind != ACT_NONE else None, gamma_a, gamma_b, ra, rb)
        ctx.cfg = (kind, act_kind, slope, b_half)
        ctx.beta_refs = (_weak(beta_a), _weak(beta_b))
        return y

    @staticmethod
    def backward(ctx, gy):
        a, b, y, gamma_a, gamma_b, ra, rb = ctx.saved_tensors
        kind, act_kind, slope, b_half = ctx.cfg
        g = gy.contiguous()
        if y is not None:
            g = _apply(_ActBwd, g, y, act_kind, slope)
        ga, gga, gba = _apply(_NormActBwd, g, a, None, gamma_a, ra, kind, ACT_NONE, 0.0)
        # the adjoint of the nearest upsample (sum over each 2x2 window), then the norm's backward at the low resolution
        g_b = _apply(_Pool2Down, g, 1.0) if b_half else g
        gb, ggb, gbb = _apply(_NormActBwd, g_b, b, None, gamma_b, rb, kind, ACT_NONE, 0.0)
        if _INPUT_GRADS_ONLY:
            gga = gba = ggb = gbb = None
        else:
            beta_a, beta_b = ctx.beta_refs[0](), ctx.beta_refs[1]()
            if _stash(gamma_a, gga):
                gga = None
            if _stash(beta_a, gba):
                gba = None
            if _stash(gamma_b, ggb):
                ggb = None
            if _stash(beta_b, gbb):
                gbb = None
        return ga, gga, gba, None, gb, ggb, gbb, None, None, None, None, None, None, None, None


class _NormActBwd(Function):
    """(gx, ggamma, gbeta) of y = act(norm(x)); differentiable w.r.t. gy, x, gamma through gx."""

    @staticmethod
    def forward(ctx, gy, x, y, gamma, rowbuf, kind, act_kind, slope):
        gy = _req(gy, "norm grad")
        N, C, HW = _nchw(x)
        rows = N * C
        gx = torch.empty_like(x)
        want_param = not _INPUT_GRADS_ONLY
        gg = torch.empty(C, dtype=torch.float32, device=x.device) if want_param else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if want_param else None
        scratch = torch.empty(8 * rows, dtype=torch.float32, device=x.device)
        check(L.sempyr_norm_bwd(kind, gy.data_ptr(), _p(y), x.data_ptr(), gamma.data_ptr(), rowbuf.data_ptr(),
                                rowbuf.data_ptr() + 4 * rows, N, C, HW, act_kind, slope, gx.data_ptr(), _p(gg), _p(gb),
                                scratch.data_ptr(), _stream()), "sempyr_norm_bwd")
        ctx.save_for_backward(gy, x, y, gamma, rowbuf)
        ctx.cfg = (kind, act_kind, slope)
        ctx.set_materialize_grads(False)
        if not want_param:
            gg = torch.zeros(0, device=x.device)
            gb = torch.zeros(0, device=x.device)
            ctx.mark_non_differentiable(gg, gb)
        return gx, gg, gb

    @staticmethod
    def backward(ctx, u, ugg, ugb):
        if ugg is not None or ugb is not None:
            raise NotImplementedError("second-order terms through the affine-parameter gradients of a norm layer "
                                      "are not part of the hot path (WGAN-GP differentiates d pred / d x only)")
        gy, x, y, gamma, rowbuf = ctx.saved_tensors
        kind, act_kind, slope = ctx.cfg
        if u is None:
            return None, None, None, None, None, None, None, None
        u = _req(u, "norm double-backward grad")
        N, C, HW = _nchw(x)
        rows = N * C
        g_gy = torch.empty_like(x)
        g_x = torch.empty_like(x)
        want_param = ctx.needs_input_grad[3] and not _INPUT_GRADS_ONLY
        g_gamma = torch.empty(C, dtype=torch.float32, device=x.device) if want_param else None
        scratch = torch.empty(19 * rows, dtype=torch.float32, device=x.device)
        check(L.sempyr_norm_bwd_bwd(kind, u.data_ptr(), gy.data_ptr(), _p(y), x.data_ptr(), gamma.data_ptr(),
                                    rowbuf.data_ptr(), rowbuf.data_ptr() + 4 * rows, N, C, HW, act_kind, slope,
                                    g_gy.data_ptr(), g_x.data_ptr(), _p(g_gamma), scratch.data_ptr(), _stream()),
              "sempyr_norm_bwd_bwd")
        if _stash(gamma, g_gamma):
            g_gamma = None
        return g_gy, g_x, None, g_gamma, None, None, None, None


def norm_act(x, gamma, beta, kind, act_kind=ACT_NONE, slope=0.0, eps=1e-5, bn_state=None):
    """Training-mode BatchNorm2d / InstanceNorm2d(affine) / GroupNorm(1,C) fused with ReLU or
    LeakyReLU (reference generator.py:27-29, discriminator.py:37-39).  `bn_state` =
    (running_mean, running_var, num_batches_tracked, momentum) is updated in place for BATCH."""
    return _apply(_NormAct, x, gamma, beta, kind, act_kind, slope, eps, bn_state, tile_stats_of(x) if kind == NORM_BATCH else None)


def norm_add_norm_act(a, gamma_a, beta_a, b, gamma_b, beta_b, kind, act_kind=ACT_NONE, slope=0.0, eps_a=1e-5, eps_b=1e-5,
                      state_a=None, state_b=None, b_half=False):
    """act(norm(a) + norm(b)) for two training-mode norm layers of the same kind (generator.py:54-57).  `b_half`: b has
    half the resolution and stands for its nearest x2 upsample (BATCH norm only: the statistics of the upsampled map are
    those of b)."""
    if b_half and kind != NORM_BATCH:
        raise _lib.SempyrError("norm_add_norm_act: a half-resolution branch is only defined for BatchNorm")
    return _apply(_NormAddNormAct, a, gamma_a, beta_a, state_a, b, gamma_b, beta_b, state_b, kind, act_kind, slope, eps_a, eps_b,
                  b_half, tile_stats_of(a) if kind == NORM_BATCH else None)


class _AffineRowAct(Function):
    """y = act(x*scale_row + shift_row [+ res]) with constant (non-differentiable) coefficients."""

    @staticmethod
    def forward(ctx, x, res, coeffs, act_kind, slope):
        x = _req(x, "input")
        if res is not None:
            res = _req(res, "residual")
        N, C, HW = _nchw(x)
        rows = N * C
        y = torch.empty_like(x)
        check(L.sempyr_affine_act(x.data_ptr(), 0, coeffs.data_ptr(), coeffs.data_ptr() + 4 * rows, _p(res), y.data_ptr(),
                                  rows, HW, act_kind, slope, _stream()), "sempyr_affine_act")
        ctx.save_for_backward(y if act_kind != ACT_NONE else None, coeffs)
        ctx.cfg = (act_kind, slope, res is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, coeffs = ctx.saved_tensors
        act_kind, slope, has_res = ctx.cfg
        if torch.is_grad_enabled() and gy.requires_grad:
            raise NotImplementedError("double backward through eval-mode BatchNorm is not part of the hot path")
        gy = _req(gy, "grad")
        N, C, HW = _nchw(gy)
        rows = N * C
        gx = torch.empty_like(gy) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(gy) if (has_res and ctx.needs_input_grad[1]) else None
        if gx is None and gres is None:
            return None, None, None, None, None
        check(L.sempyr_affine_act_bwd(gy.data_ptr(), _p(y), coeffs.data_ptr(), _p(gx), _p(gres), rows, HW, act_kind, slope,
                                      _stream()), "sempyr_affine_act_bwd")
        return gx, gres, None, None, None


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, N):
    C = gamma.numel()
    coeffs = torch.empty(2 * N * C, dtype=torch.float32, device=gamma.device)
    check(L.sempyr_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps,
                                  N, C, coeffs.data_ptr(), coeffs.data_ptr() + 4 * N * C, _stream()), "sempyr_bn_eval_coeffs")
    return coeffs


def _bn_eval_cached(gamma, beta, running_mean, running_var, eps, N):
    """Per-row (N*C) scale / shift of an eval-mode BatchNorm2d.  They only change when a parameter / buffer does, so they
    are cached on the weight tensor (keyed by batch size and tensor versions): the frozen classifier computes them once."""
    key = (N, eps, gamma._version, beta._version, running_mean._version, running_var._version,
           gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr())
    hit = getattr(gamma, "_sempyr_bn_eval", None)
    if hit is not None and hit[0] == key:
        coeffs = hit[1]
    else:
        coeffs = bn_eval_coeffs(_req(gamma), _req(beta), _req(running_mean), _req(running_var), eps, N)
        try:
            gamma._sempyr_bn_eval = (key, coeffs)
        except Exception:  # pragma: no cover
            pass
    return coeffs


def bn_eval_act(x, gamma, beta, running_mean, running_var, eps=1e-5, res=None, act_kind=ACT_NONE, slope=0.0):
    """Eval-mode BatchNorm2d (+ residual add + activation) with frozen statistics and parameters
    (ResNet18 classifier, classifier.py:34 under .eval(); generator under `sample`, train_model.py:153)."""
    coeffs = _bn_eval_cached(gamma, beta, running_mean, running_var, eps, x.shape[0])
    return _apply(_AffineRowAct, x, res, coeffs, act_kind, slope)


class _ConvBnEval(Function):
    """y = act(BatchNorm_eval(conv2d(x[, upsampled x2], w)) [+ res]) in one launch: the norm's per-channel scale / shift,
    the residual add and the activation run in the convolution's epilogue.  `coeffs` are constants, as in _AffineRowAct."""

    @staticmethod
    def forward(ctx, x, w, res, plan, coeffs, act_kind, slope, res_half=False):
        x = _req(x, "conv input"); w = _req(w, "conv weight")
        if res is not None:
            res = _req(res, "residual")
            want = (plan.B, plan.Cout, plan.Ho // 2, plan.Wo // 2) if res_half else (plan.B, plan.Cout, plan.Ho, plan.Wo)
            if tuple(res.shape) != want or (res_half and (plan.Ho % 2 or plan.Wo % 2)):
                raise _lib.SempyrError("conv + BatchNorm: the residual must have the output's shape (or half of it)")
        elif res_half:
            raise _lib.SempyrError("conv + BatchNorm: res_half without a residual")
        rows = plan.B * plan.Cout
        if coeffs.numel() != 2 * rows:
            raise _lib.SempyrError("conv + BatchNorm: coefficient table does not match the output")
        y = _raw_conv_fwd(plan, x, w, coeffs.data_ptr() + 4 * rows, act_kind, slope, None, coeffs.data_ptr(), res, res_half)
        ctx.save_for_backward(x, w, y if act_kind != ACT_NONE else None, coeffs)
        ctx.plan = plan
        ctx.cfg = (act_kind, slope, res is not None, res_half)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, coeffs = ctx.saved_tensors
        plan = ctx.plan
        act_kind, slope, has_res, res_half = ctx.cfg
        if torch.is_grad_enabled() and gy.requires_grad:
            raise NotImplementedError("double backward through eval-mode BatchNorm is not part of the hot path")
        gy = _req(gy, "grad")
        rows, HW = plan.B * plan.Cout, plan.Ho * plan.Wo
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not _INPUT_GRADS_ONLY
        gres = torch.empty_like(gy) if (has_res and ctx.needs_input_grad[2]) else None
        g = torch.empty_like(gy) if (want_x or want_w) else None
        if g is None and gres is None:
            return None, None, None, None, None, None, None, None
        check(L.sempyr_affine_act_bwd(gy.data_ptr(), _p(y), coeffs.data_ptr(), _p(g), _p(gres), rows, HW, act_kind, slope,
                                      _stream()), "sempyr_affine_act_bwd")
        if gres is not None and res_half:
            gres = _apply(_Pool2Down, gres, 1.0)          # adjoint of the nearest upsample the residual was read through
        gx = gw = None
        if want_w:
            if _can_async(w, None, True, False):
                _param_grads_async(g, x, plan, w, None, True, False)
            else:
                gw = _apply(_ConvWgrad, g, x, plan)
        if want_x:
            gx = _conv_input_grad(g, w, plan)
        return gx, gw, gres, None, None, None, None, None


def conv_bn_eval_act(x, w, gamma, beta, running_mean, running_var, eps=1e-5, stride=1, padding=0, upsample=False, res=None,
                     act_kind=ACT_NONE, slope=0.0, res_half=False):
    """Bias-free nn.Conv2d followed by an eval-mode nn.BatchNorm2d, an optional residual add and an activation — the
    BasicBlock pattern of the frozen ResNet18 classifier (classifier.py:34-40) and the generator's ConvBlock / ResidualBlock
    under .eval() (generator.py:27-29,54-57 from eval_model.py) — as a single convolution launch."""
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = w.shape
    if Cin_w != Cin:
        raise _lib.SempyrError(f"conv2d: weight expects {Cin_w} input channels, got {Cin}")
    if gamma.numel() != Cout:
        raise _lib.SempyrError(f"BatchNorm2d over {gamma.numel()} channels after a conv with {Cout} outputs")
    plan = get_plan((B, Cin, H, W, Cout, kh, kw, stride, padding, 1 if upsample else 0), x.device)
    coeffs = _bn_eval_cached(gamma, beta, running_mean, running_var, eps, B)
    return _apply(_ConvBnEval, x, w, res, plan, coeffs, act_kind, slope, res_half)


# ------------------------------------------------------------------------------------------------
# activations / residual add
# ------------------------------------------------------------------------------------------------
def _act_raw(x, res, act_kind, slope):
    y = torch.empty_like(x)
    check(L.sempyr_affine_act(x.data_ptr(), 0, 0, 0, _p(res), y.data_ptr(), 1, x.numel(), act_kind, slope, _stream()),
          "sempyr_affine_act")
    return y


class _AddAct(Function):
    """y = act(a [+ b])"""

    @staticmethod
    def forward(ctx, a, b, act_kind, slope):
        a = _req(a, "input")
        if b is not None:
            b = _req(b, "input")
            if b.shape != a.shape:
                raise _lib.SempyrError("add_act: shape mismatch")
        y = _act_raw(a, b, act_kind, slope)
        ctx.save_for_backward(y if act_kind != ACT_NONE else None)
        ctx.cfg = (act_kind, slope, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        act_kind, slope, has_b = ctx.cfg
        g = gy if act_kind == ACT_NONE else _apply(_ActBwd, gy, y, act_kind, slope)
        return g, (g if has_b else None), None, None


class _ActBwd(Function):
    """g = gy * act'(y) (linear in gy)."""

    @staticmethod
    def forward(ctx, gy, y, act_kind, slope):
        gy = _req(gy, "grad")
        g = torch.empty_like(gy)
        check(L.sempyr_affine_act_bwd(gy.data_ptr(), y.data_ptr(), 0, g.data_ptr(), 0, 1, gy.numel(), act_kind, slope,
                                      _stream()), "sempyr_affine_act_bwd")
        ctx.save_for_backward(y)
        ctx.cfg = (act_kind, slope)
        return g

    @staticmethod
    def backward(ctx, u):
        (y,) = ctx.saved_tensors
        act_kind, slope = ctx.cfg
        if act_kind == ACT_TANH:
            raise NotImplementedError("double backward through tanh is not part of the hot path")
        return _apply(_ActBwd, u.contiguous(), y, act_kind, slope), None, None, None


def act(x, act_kind, slope=0.0):
    """nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh (generator.py:48,120 ; discriminator.py:99,102)."""
    return _apply(_AddAct, x, None, act_kind, slope)


def add_act(a, b, act_kind=ACT_NONE, slope=0.0):
    """`x += residual; x = activation(x)` (generator.py:56-57, discriminator.py:63-64) and gen + features
    (generator.py:87)."""
    return _apply(_AddAct, a, b, act_kind, slope)


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _req(a, "input"); b = _req(b, "input")
        if a.shape != b.shape:
            raise _lib.SempyrError("mul: shape mismatch")
        ctx.save_for_backward(a, b)
        y = torch.empty_like(a)
        check(L.sempyr_mul(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "sempyr_mul")
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _apply(_Mul, g.contiguous(), b) if ctx.needs_input_grad[0] else None
        gb = _apply(_Mul, g.contiguous(), a) if ctx.needs_input_grad[1] else None
        return ga, gb


def mul(a, b):
    """features[i] * masks[i] (generator.py:126-129)."""
    return _apply(_Mul, a, b)


class _Normalize(Function):
    @staticmethod
    def forward(ctx, x, mean_c, std_c, backward_mode):
        x = _req(x, "input")
        N, C, HW = _nchw(x)
        y = torch.empty_like(x)
        check(L.sempyr_normalize(x.data_ptr(), _p(mean_c), std_c.data_ptr(), y.data_ptr(), N, C, HW,
                                 1 if backward_mode else 0, _stream()), "sempyr_normalize")
        ctx.save_for_backward(mean_c, std_c)
        return y

    @staticmethod
    def backward(ctx, g):
        mean_c, std_c = ctx.saved_tensors
        return _apply(_Normalize, g.contiguous(), mean_c, std_c, True), None, None, None


def normalize(x, mean_c, std_c):
    """transforms.Normalize(mean, std) on a batch (train_model.py:240-245,308-309,98)."""
    return _apply(_Normalize, x, mean_c, std_c, False)


# ------------------------------------------------------------------------------------------------
# pooling / upsampling
# ------------------------------------------------------------------------------------------------
class _Pool2Down(Function):
    """y = scale * sum_2x2 (x * act'(yref)); yref = None: plain 2x2 sum."""

    @staticmethod
    def forward(ctx, x, scale, yref=None, act_kind=ACT_NONE, slope=0.0, bits=None):
        x = _req(x, "input")
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(L.sempyr_pool2_down(x.data_ptr(), _p(yref), _p(bits), y.data_ptr(), N * C, H, W, scale, act_kind, slope, _stream()),
              "sempyr_pool2_down")
        ctx.save_for_backward(yref, bits)
        ctx.cfg = (scale, act_kind, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        yref, bits = ctx.saved_tensors
        scale, act_kind, slope = ctx.cfg
        return _apply(_Pool2Up, g.contiguous(), scale, yref, act_kind, slope, bits), None, None, None, None, None


def _pool2_up_raw(x, scale, yref=None, act_kind=ACT_NONE, slope=0.0, partials=None, bits=None):
    """`partials` ([N*C, h*w/512] floats): also receive per-row partial sums of the result (see sempyr_pool2_up).
    `bits`: the sign bits of the fused stem convolution, instead of yref."""
    N, C, h, w = x.shape
    y = torch.empty((N, C, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    check(L.sempyr_pool2_up(x.data_ptr(), _p(yref), _p(bits), y.data_ptr(), N * C, h, w, scale, act_kind, slope, _p(partials),
                            _stream()), "sempyr_pool2_up")
    return y


class _Pool2Up(Function):
    """y = scale * nearest_up2(x) * act'(yref); yref = None: plain nearest upsample."""

    @staticmethod
    def forward(ctx, x, scale, yref=None, act_kind=ACT_NONE, slope=0.0, bits=None):
        y = _pool2_up_raw(_req(x, "input"), scale, yref, act_kind, slope, bits=bits)
        ctx.save_for_backward(yref, bits)
        ctx.cfg = (scale, act_kind, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        yref, bits = ctx.saved_tensors
        scale, act_kind, slope = ctx.cfg
        return _apply(_Pool2Down, g.contiguous(), scale, yref, act_kind, slope, bits), None, None, None, None, None


def avg_pool2(x):
    """nn.AvgPool2d(2) (discriminator.py:57,103)."""
    return _apply(_Pool2Down, x, 0.25)


def upsample2(x):
    """nn.UpsamplingNearest2d(scale_factor=2) as a standalone op (normally fused into conv2d)."""
    return _apply(_Pool2Up, x, 1.0)


class _MaxPool3s2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "input")
        N, C, H, W = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, C, Ho, Wo), dtype=torch.int32, device=x.device)
        check(L.sempyr_maxpool3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N * C, H, W, _stream()),
              "sempyr_maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.in_shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled() and g.requires_grad:
            raise NotImplementedError("double backward through max-pool is not part of the hot path")
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.in_shape
        g = _req(g, "grad")
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        check(L.sempyr_maxpool3s2_bwd(g.data_ptr(), idx.data_ptr(), gx.data_ptr(), N * C, H, W, _stream()),
              "sempyr_maxpool3s2_bwd")
        return gx


def max_pool3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet18 stem (classifier.py:36)."""
    return _apply(_MaxPool3s2, x)


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d((1,1)) (classifier.py:41).  Forward only: tap 5 never enters a loss
    (train_model.py:212-222 uses taps 1..4)."""
    x = _req(x.detach(), "input")
    N, C, HW = _nchw(x)
    out = torch.empty(N * C, dtype=torch.float32, device=x.device)
    scratch = torch.empty(N * C, dtype=torch.float32, device=x.device)
    # channel_sum over a (1, N*C, HW) view = per-row sums
    check(L.sempyr_channel_sum(x.data_ptr(), 1, N * C, HW, 1.0 / HW, out.data_ptr(), 0, scratch.data_ptr(), _stream()),
          "sempyr_channel_sum")
    return out.view(N, C, 1, 1)


# ------------------------------------------------------------------------------------------------
# per-iteration scalars in device memory (inputs of captured graphs) and the feature masks
# ------------------------------------------------------------------------------------------------
def store_words(dst, words):
    """dst[0:len(words)] = words (Python ints, taken modulo 2**32), stream-ordered; dst is an int32 / float32 device tensor.
    The words travel in the kernel arguments (no pinned staging buffer whose reuse would need an event)."""
    n = len(words)
    if n > _lib.STORE_MAX_WORDS or dst.numel() < n or dst.element_size() != 4 or not dst.is_cuda:
        raise _lib.SempyrError("store_words: at most %d words into a 4-byte device tensor" % _lib.STORE_MAX_WORDS)
    arr = (ctypes.c_uint32 * n)(*[w & 0xFFFFFFFF for w in words])
    check(L.sempyr_store_words(arr, n, dst.data_ptr(), _raw_stream(dst.device.index)), "sempyr_store_words")


def float_bits(v):
    import struct
    return struct.unpack('<I', struct.pack('<f', v))[0]


def build_masks(masks, spec_dev):
    """Fill the fp32 (N, C, H, W) tensors `masks` from their six-word specs {bits(base), bits(win), x0, x1, y0, y1} held in
    the int32 device tensor `spec_dev` (train_model.py:172-209), one launch for all of them."""
    n = len(masks)
    arr = (_lib.MaskTarget * n)()
    for i, m in enumerate(masks):
        _req(m, "mask")
        if m.dim() != 4 or not m.is_contiguous():
            raise _lib.SempyrError("build_masks: contiguous (N, C, H, W) tensors expected")
        arr[i].mask = m.data_ptr(); arr[i].rows = m.shape[0] * m.shape[1]; arr[i].H = m.shape[2]; arr[i].W = m.shape[3]
    if spec_dev.dtype != torch.int32 or spec_dev.numel() < 6 * n:
        raise _lib.SempyrError("build_masks: the spec is an int32 tensor of 6 words per mask")
    check(L.sempyr_build_masks(arr, n, spec_dev.data_ptr(), _raw_stream(masks[0].device.index)), "sempyr_build_masks")


# ------------------------------------------------------------------------------------------------
# loss-path reductions
# ------------------------------------------------------------------------------------------------
class _Mean(Function):
    """scale * mean(x) -> 0-d tensor."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _req(x, "input")
        out = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty(_REDUCE_SCRATCH, dtype=torch.float32, device=x.device)
        check(L.sempyr_mean(x.data_ptr(), x.numel(), scale, out.data_ptr(), scratch.data_ptr(), _stream()), "sempyr_mean")
        ctx.shape = x.shape
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled() and g.requires_grad:
            raise NotImplementedError("double backward through a mean is not part of the hot path")
        g = _req(g, "grad")
        n = 1
        for s in ctx.shape:
            n *= s
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(L.sempyr_fill_scalar(g.data_ptr(), ctx.scale / n, gx.data_ptr(), n, _stream()), "sempyr_fill_scalar")
        return gx, None


def mean(x, scale=1.0):
    """`x.mean()` / `-x.mean()` of the WGAN losses (train_model.py:39-46)."""
    return _apply(_Mean, x, scale)


class _L1Mean(Function):
    """mean |a*m - b*m| (m optional)."""

    @staticmethod
    def forward(ctx, a, b, m):
        a = _req(a, "input"); b = _req(b, "input")
        if a.shape != b.shape or (m is not None and m.shape != a.shape):
            raise _lib.SempyrError("l1_mean: shape mismatch")
        if m is not None:
            m = _req(m, "mask")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        scratch = torch.empty(_REDUCE_SCRATCH, dtype=torch.float32, device=a.device)
        check(L.sempyr_l1_mean(a.data_ptr(), b.data_ptr(), _p(m), a.numel(), out.data_ptr(), scratch.data_ptr(), _stream()),
              "sempyr_l1_mean")
        ctx.save_for_backward(a, b, m)
        return out

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled() and g.requires_grad:
            raise NotImplementedError("double backward through the L1 loss is not part of the hot path")
        a, b, m = ctx.saved_tensors
        g = _req(g, "grad")
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if ga is None and gb is None:
            return None, None, None
        check(L.sempyr_l1_mean_bwd(a.data_ptr(), b.data_ptr(), _p(m), g.data_ptr(), a.numel(), _p(ga), _p(gb), _stream()),
              "sempyr_l1_mean_bwd")
        return ga, gb, None


def l1_mean(a, b, mask=None):
    """nn.L1Loss()(a*mask, b*mask) in one pass (train_model.py:217-221) ; mask=None: nn.L1Loss()(a, b)
    (train_model.py:109 ; train_feature_creation_gen.py:47,50)."""
    return _apply(_L1Mean, a, b, mask)


class _GPLoss(Function):
    """mean_b (||grad_b||_2 - 1)^2 (train_model.py:58-66)."""

    @staticmethod
    def forward(ctx, grad):
        grad = _req(grad, "gradient")
        rows = grad.shape[0]
        n = grad.numel() // rows
        norms = torch.empty(rows, dtype=torch.float32, device=grad.device)
        gp = torch.empty((), dtype=torch.float32, device=grad.device)
        scratch = torch.empty(rows * 48, dtype=torch.float32, device=grad.device)       # SEMPYR_GP_SPLIT partials per row
        check(L.sempyr_gp_fwd(grad.data_ptr(), rows, n, norms.data_ptr(), gp.data_ptr(), scratch.data_ptr(), _stream()),
              "sempyr_gp_fwd")
        ctx.save_for_backward(grad, norms)
        return gp

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled() and g.requires_grad:
            raise NotImplementedError("third-order derivatives are not part of the hot path")
        grad, norms = ctx.saved_tensors
        g = _req(g, "grad")
        rows = grad.shape[0]
        out = torch.empty_like(grad)
        check(L.sempyr_gp_bwd(grad.data_ptr(), norms.data_ptr(), g.data_ptr(), rows, grad.numel() // rows, out.data_ptr(),
                              _stream()), "sempyr_gp_bwd")
        return out


def gp_loss(grad):
    """one_mean_gp(grad) (train_model.py:63-66)."""
    return _apply(_GPLoss, grad)


class _ConcatRows(Function):
    """cat(tensors, dim=0) in one launch; the backward hands out views of the incoming gradient."""

    @staticmethod
    def forward(ctx, *tensors):
        ts = [_req(t, "concat input") for t in tensors]
        tail = tuple(ts[0].shape[1:])
        if len(ts) > _lib.CONCAT_MAX or any(tuple(t.shape[1:]) != tail for t in ts):
            raise _lib.SempyrError("concat_rows: up to %d tensors of equal trailing shape" % _lib.CONCAT_MAX)
        rows = [t.shape[0] for t in ts]
        out = torch.empty((sum(rows),) + tail, dtype=torch.float32, device=ts[0].device)
        srcs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ns = (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])
        check(L.sempyr_concat(srcs, ns, len(ts), out.data_ptr(), _stream()), "sempyr_concat")
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for i, r in enumerate(ctx.rows):
            outs.append(g[off:off + r] if ctx.needs_input_grad[i] else None)
            off += r
        return tuple(outs)


def concat_rows(tensors):
    """torch.cat(tensors, dim=0) — the real and the fake batch as one critic batch (train_model.py:132-133)."""
    return _apply(_ConcatRows, *tensors)


def gather_scalars(scalars, out_row):
    """out_row[i] = scalars[i] for up to 8 one-element device tensors, one launch."""
    n = len(scalars)
    if n > _lib.CONCAT_MAX or out_row.numel() < n or not out_row.is_contiguous():
        raise _lib.SempyrError("gather_scalars: up to %d scalars into a contiguous row" % _lib.CONCAT_MAX)
    srcs = (ctypes.c_void_p * n)(*[_req(v, "scalar").data_ptr() for v in scalars])
    ns = (ctypes.c_int64 * n)(*([1] * n))
    check(L.sempyr_concat(srcs, ns, n, out_row.data_ptr(), _stream()), "sempyr_concat")


def full_like(t, value):
    """torch.full_like(t, value) (grad_outputs=torch.ones_like(pred), train_model.py:73)."""
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    check(L.sempyr_fill_const(out.data_ptr(), float(value), out.numel(), _stream()), "sempyr_fill_const")
    return out


def _lincomb_raw(xs, ws, s1, s2):
    n = len(xs)
    out = torch.empty((), dtype=torch.float32, device=xs[0].device)
    px = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    pw = (ctypes.c_float * n)(*ws)
    check(L.sempyr_lincomb(px, pw, n, s1, s2, out.data_ptr(), _stream()), "sempyr_lincomb")
    return out


class _Lincomb(Function):
    """((w0*x0 + w1*x1 + ...) * s1) * s2 on 0-d tensors."""

    @staticmethod
    def forward(ctx, ws, s1, s2, *xs):
        xs = [_req(x, "loss term") for x in xs]
        if any(x.numel() != 1 for x in xs) or len(xs) != len(ws) or len(xs) > _lib.LINCOMB_MAX:
            raise _lib.SempyrError("lincomb: up to %d one-element tensors, one weight each" % _lib.LINCOMB_MAX)
        ctx.cfg = (tuple(float(w) for w in ws), float(s1), float(s2))
        return _lincomb_raw(xs, ctx.cfg[0], ctx.cfg[1], ctx.cfg[2])

    @staticmethod
    def backward(ctx, g):
        ws, s1, s2 = ctx.cfg
        if torch.is_grad_enabled() and g.requires_grad:
            raise NotImplementedError("double backward through the loss arithmetic is not part of the hot path")
        g = _req(g, "grad")
        outs = []
        cache = {}
        for i, w in enumerate(ws):
            if not ctx.needs_input_grad[3 + i]:
                outs.append(None)
                continue
            if w == 1.0 and s1 == 1.0 and s2 == 1.0:
                outs.append(g)
                continue
            if w not in cache:
                cache[w] = _lincomb_raw([g], [s2], s1, w)          # ((g*s2)*s1)*w : the order of the chain of torch ops
            outs.append(cache[w])
        return (None, None, None) + tuple(outs)


def lincomb(xs, ws=None, s1=1.0, s2=1.0):
    """The scalar loss arithmetic of the step functions (train_model.py:101-103,112,138,222) in one launch."""
    ws = [1.0] * len(xs) if ws is None else ws
    return _apply(_Lincomb, tuple(ws), s1, s2, *xs)


class _MeanParts(Function):
    """(scale_i * mean(x[start_i : start_i + rows_i]))_i for consecutive row blocks of x."""

    @staticmethod
    def forward(ctx, x, rows, scales):
        x = _req(x, "input")
        if sum(rows) != x.shape[0] or len(rows) != len(scales):
            raise _lib.SempyrError("mean_parts: the row blocks must cover the tensor")
        per = x.numel() // x.shape[0]
        outs, off = [], 0
        for r, sc in zip(rows, scales):
            out = torch.empty((), dtype=torch.float32, device=x.device)
            scratch = torch.empty(_REDUCE_SCRATCH, dtype=torch.float32, device=x.device)
            check(L.sempyr_mean(x.data_ptr() + 4 * off * per, r * per, sc, out.data_ptr(), scratch.data_ptr(), _stream()), "sempyr_mean")
            outs.append(out)
            off += r
        ctx.cfg = (tuple(rows), tuple(scales), tuple(x.shape), per)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        rows, scales, shape, per = ctx.cfg
        if any(g is not None and torch.is_grad_enabled() and g.requires_grad for g in gs):
            raise NotImplementedError("double backward through a mean is not part of the hot path")
        dev = next(g for g in gs if g is not None).device
        gx = torch.empty(shape, dtype=torch.float32, device=dev)
        off = 0
        for r, sc, g in zip(rows, scales, gs):
            dst = gx.data_ptr() + 4 * off * per
            if g is None:
                check(L.sempyr_fill_const(dst, 0.0, r * per, _stream()), "sempyr_fill_const")
            else:
                check(L.sempyr_fill_scalar(_req(g, "grad").data_ptr(), sc / (r * per), dst, r * per, _stream()), "sempyr_fill_scalar")
            off += r
        return gx, None, None


def mean_parts(x, rows, scales):
    """The WGAN critic losses (-mean(real logits), mean(fake logits)) of train_model.py:39-42 taken from ONE logits tensor
    of the batched critic pass, without slicing it (a slice's backward is a fill + copy + add in torch)."""
    return _apply(_MeanParts, x, tuple(rows), tuple(scales))


def lerp_rows(real, fake, alpha):
    """real + alpha*(fake-real) with one alpha per sample (sample_line, train_model.py:51-55); the result
    is detached by its only caller (train_model.py:70)."""
    real = _req(real.detach(), "real"); fake = _req(fake.detach(), "fake")
    if real.shape != fake.shape:
        raise _lib.SempyrError("lerp_rows: shape mismatch")
    alpha = _req(alpha.reshape(-1), "alpha")
    rows = real.shape[0]
    if alpha.numel() != rows:
        raise _lib.SempyrError("lerp_rows: one alpha per sample expected")
    out = torch.empty_like(real)
    check(L.sempyr_lerp_rows(real.data_ptr(), fake.data_ptr(), alpha.data_ptr(), out.data_ptr(), rows, real.numel() // rows,
                             _stream()), "sempyr_lerp_rows")
    return out
