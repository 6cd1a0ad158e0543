"""Building blocks shared by the generator / discriminator / classifier mirrors.

The `torch.nn` layer objects (Conv2d, ConvTranspose2d, BatchNorm2d, InstanceNorm2d, GroupNorm) are
kept purely as *parameter containers*: they give the reference's state_dict keys, dtypes and shapes
 and consume the RNG exactly like the reference at construction time, but their own
`forward` is never called — the runners below execute the hand-written HIP operators instead.
"""
import torch.nn as nn

from . import ops


def init_weights(m):
    """DCGAN-style initialisation, selected by class name exactly as the reference does
    (generator.py:4-12, discriminator.py:5-13): N(0, .02) for Conv*/Linear weights, N(1, .02) / 0 for
    BatchNorm weight / bias.  (InstanceNorm2d and GroupNorm keep their default ones/zeros.)"""
    name = type(m).__name__
    if 'Conv' in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif 'BatchNorm' in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)
    elif 'Linear' in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)


def _single(v):
    return v[0] if isinstance(v, (tuple, list)) else v


def run_conv(layer, x, upsample=False, act_kind=ops.ACT_NONE, slope=0.0, pool=False, stats=False):
    """Execute an nn.Conv2d / nn.ConvTranspose2d container on the HIP implicit-GEMM path; an activation that
    directly follows a Conv2d runs in the kernel epilogue.  `stats`: a training-mode BatchNorm2d consumes the result
    (see ops.conv2d)."""
    stride, pad = _single(layer.stride), _single(layer.padding)
    if isinstance(layer, nn.ConvTranspose2d):
        if upsample:
            raise ValueError("fused upsample is only defined for Conv2d")
        y = ops.conv_transpose2d(x, layer.weight, layer.bias, stride, pad)
        y = ops.act(y, act_kind, slope) if act_kind != ops.ACT_NONE else y
        return ops.avg_pool2(y) if pool else y
    return ops.conv2d(x, layer.weight, layer.bias, stride, pad, upsample=upsample, act_kind=act_kind, slope=slope, pool=pool,
                      stats=stats)


def run_norm(layer, x, act_kind=ops.ACT_NONE, slope=0.0, res=None):
    """Execute a norm container (+ fused activation).  `res` (a residual added before the activation)
    is only available for eval-mode BatchNorm, where the whole thing is one affine pass."""
    if isinstance(layer, nn.Identity):
        if res is not None:
            return ops.add_act(x, res, act_kind, slope)
        return ops.act(x, act_kind, slope) if act_kind != ops.ACT_NONE else x
    if isinstance(layer, nn.BatchNorm2d):
        if layer.training or not layer.track_running_stats:
            state = None
            if layer.track_running_stats:
                if layer.momentum is None:
                    raise NotImplementedError("cumulative-average BatchNorm momentum is not used by the reference")
                state = (layer.running_mean, layer.running_var, layer.num_batches_tracked, float(layer.momentum))
            y = ops.norm_act(x, layer.weight, layer.bias, ops.NORM_BATCH, act_kind if res is None else ops.ACT_NONE,
                             slope, layer.eps, state)
            return ops.add_act(y, res, act_kind, slope) if res is not None else y
        return ops.bn_eval_act(x, layer.weight, layer.bias, layer.running_mean, layer.running_var, layer.eps,
                               res=res, act_kind=act_kind, slope=slope)
    if isinstance(layer, nn.InstanceNorm2d):
        if layer.track_running_stats or not layer.affine:
            raise NotImplementedError("only InstanceNorm2d(affine=True, track_running_stats=False) is on the hot path")
        kind = ops.NORM_INSTANCE
    elif isinstance(layer, nn.GroupNorm):
        if layer.num_groups != 1:
            raise NotImplementedError("only GroupNorm(1, C) ('layer_norm') is on the hot path")
        kind = ops.NORM_LAYER
    else:
        raise NotImplementedError(f"no HIP kernel for {type(layer).__name__}")
    y = ops.norm_act(x, layer.weight, layer.bias, kind, act_kind if res is None else ops.ACT_NONE, slope, layer.eps)
    return ops.add_act(y, res, act_kind, slope) if res is not None else y


def fuses_conv_norm(conv, norm):
    """True when run_conv_norm runs `conv` + `norm` as one launch (bias-free Conv2d + eval-mode BatchNorm2d)."""
    return (isinstance(conv, nn.Conv2d) and conv.bias is None and conv.out_channels > 4 and isinstance(norm, nn.BatchNorm2d)
            and not norm.training and norm.track_running_stats and norm.affine)


def run_conv_norm(conv, norm, x, act_kind=ops.ACT_NONE, slope=0.0, res=None, upsample=False, res_half=False):
    """act(norm(conv(x)) [+ res]).  A bias-free Conv2d followed by an eval-mode BatchNorm2d (every conv of the frozen
    ResNet18 classifier; the generator under .eval()) is one launch: scale / shift, residual and activation run in the
    convolution's epilogue.  Everything else is run_conv followed by run_norm."""
    if fuses_conv_norm(conv, norm):
        return ops.conv_bn_eval_act(x, conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps,
                                    _single(conv.stride), _single(conv.padding), upsample=upsample, res=res,
                                    act_kind=act_kind, slope=slope, res_half=res_half)
    if res_half:
        raise NotImplementedError("a half-resolution residual needs the fused conv + eval-BatchNorm path (see fuses_conv_norm)")
    k = _train_norm_kind(norm)
    return run_norm(norm, run_conv(conv, x, upsample=upsample, stats=k is not None and k[0] == ops.NORM_BATCH), act_kind, slope,
                    res=res)


def _train_norm_kind(layer):
    """(kind, bn_state) when `layer` is a norm container that computes batch statistics right now, else None."""
    if isinstance(layer, nn.BatchNorm2d):
        if not (layer.training or not layer.track_running_stats):
            return None
        state = None
        if layer.track_running_stats:
            if layer.momentum is None:
                return None
            state = (layer.running_mean, layer.running_var, layer.num_batches_tracked, float(layer.momentum))
        return ops.NORM_BATCH, state
    if isinstance(layer, nn.InstanceNorm2d) and layer.affine and not layer.track_running_stats:
        return ops.NORM_INSTANCE, None
    if isinstance(layer, nn.GroupNorm) and layer.num_groups == 1:
        return ops.NORM_LAYER, None
    return None


def tail_takes_half_res(norm_a, norm_b):
    """True when run_norm_add_norm_act(..., b_half=True) is available: two training-mode BatchNorms."""
    ka, kb = _train_norm_kind(norm_a), _train_norm_kind(norm_b)
    return ka is not None and kb is not None and ka[0] == kb[0] == ops.NORM_BATCH


def run_norm_add_norm_act(norm_a, a, norm_b, b, act_kind=ops.ACT_NONE, slope=0.0, b_half=False):
    """act(norm_a(a) + norm_b(b)) — the tail of a residual block.  Two training-mode norms of the same kind run as one
    fused apply pass; anything else falls back to norm_b, then norm_a with the residual.  `b_half`: b is the skip branch
    taken BEFORE the block's nearest x2 upsample (see ops.norm_add_norm_act; check tail_takes_half_res first)."""
    ka, kb = _train_norm_kind(norm_a), _train_norm_kind(norm_b)
    if ka is not None and kb is not None and ka[0] == kb[0]:
        return ops.norm_add_norm_act(a, norm_a.weight, norm_a.bias, b, norm_b.weight, norm_b.bias, ka[0], act_kind, slope,
                                     norm_a.eps, norm_b.eps, ka[1], kb[1], b_half=b_half)
    if b_half:
        raise NotImplementedError("a half-resolution skip branch needs two training-mode BatchNorm layers")
    return run_norm(norm_a, a, act_kind, slope, res=run_norm(norm_b, b))


def activation_kind(layer):
    """Map an activation container to the fused-activation code."""
    if isinstance(layer, nn.ReLU):
        return ops.ACT_RELU, 0.0
    if isinstance(layer, nn.LeakyReLU):
        return ops.ACT_LRELU, float(layer.negative_slope)
    if isinstance(layer, nn.Tanh):
        return ops.ACT_TANH, 0.0
    if isinstance(layer, nn.Identity):
        return ops.ACT_NONE, 0.0
    raise NotImplementedError(f"no HIP kernel for activation {type(layer).__name__}")


def run_conv_norm_act(block, x, upsample=False):
    """conv -> norm -> activation for an nn.Sequential(conv, norm, act) container
    (generator.py:27-29, discriminator.py:37-39)."""
    conv, norm, act = block[0], block[1], block[2]
    kind, slope = activation_kind(act)
    return run_conv_norm(conv, norm, x, kind, slope, upsample=upsample)
