"""Operator-level parity: every C-ABI kernel family against a plain torch fp32 CPU reference of the same
op (tolerances stated per test).  These call through libsempyr_hip.so via the autograd Functions."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from semantic_pyramid_for_image_generation_amd import ops
    return ops


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# (B, Cin, H, W, Cout, k, stride, pad, ups, bias)
CONV_CASES = [
    (2, 3, 32, 32, 64, 3, 1, 1, 0, True),       # D stem (res)
    (2, 3, 32, 32, 64, 4, 2, 1, 0, True),       # D stem (default)
    (2, 64, 32, 32, 64, 4, 2, 1, 0, True),      # D block
    (2, 128, 8, 8, 256, 4, 2, 1, 0, True),
    (3, 512, 4, 4, 1, 4, 1, 0, 0, True),        # D last conv (M=1, split-K)
    (2, 64, 16, 16, 64, 3, 1, 1, 1, False),     # G residual conv with fused upsample
    (2, 64, 16, 16, 64, 1, 1, 0, 1, False),     # G skip 1x1 with fused upsample
    (2, 128, 8, 8, 64, 3, 1, 1, 0, False),
    (2, 64, 32, 32, 3, 3, 1, 1, 0, True),       # G last conv (M=3)
    (2, 3, 64, 64, 64, 7, 2, 3, 0, False),      # ResNet stem
    (2, 64, 16, 16, 128, 3, 2, 1, 0, False),    # ResNet stage entry
    (2, 64, 16, 16, 128, 1, 2, 0, 0, False),    # ResNet shortcut
    (1, 5, 9, 11, 7, 3, 1, 1, 0, True),         # ragged everything
    (2, 96, 6, 6, 96, 3, 1, 1, 0, False),       # Mpad = 96 -> 32-wide tiles
    (1, 512, 8, 8, 512, 3, 1, 1, 0, False),     # deep K, split-K path
    (2, 64, 64, 64, 3, 3, 1, 1, 0, True),       # G last conv, large enough for the 4-pixels-per-lane VALU kernel
    (2, 3, 64, 64, 64, 3, 1, 1, 0, True),       # D stem (res): its input gradient runs on the same kernel
    (1, 5, 64, 68, 3, 3, 1, 1, 0, True),        # odd channel count, non-square image
    (1, 6, 64, 64, 4, 3, 1, 1, 0, False),       # 4 output rows
    (1, 8, 16, 256, 3, 3, 1, 1, 0, True),       # width 256: one wave per image row, neighbour pixels by lane shifts
    (1, 3, 16, 256, 8, 3, 1, 1, 0, True),       # ... and as an input gradient
    (1, 8, 40, 128, 4, 3, 1, 1, 0, True),       # VALU weight-gradient kernel: 4 output rows, ragged last row strip
    (2, 4, 70, 256, 3, 3, 1, 1, 0, True),       # ... rolling-window path (width 256), three strips
    (2, 16, 24, 128, 64, 3, 1, 1, 0, False),    # patch kernel on 2-D tile blocks (rows x 64 columns of a wider image)
    (1, 16, 12, 64, 64, 3, 1, 1, 1, False),     # ... through the fused upsample (24 x 128 output)
    (1, 8, 16, 256, 32, 3, 1, 1, 0, True),      # ... four blocks per image row, bias
    (1, 64, 8, 192, 64, 3, 1, 1, 0, False),     # ... width 192 = 3 blocks
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, case):
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, ups, has_bias = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) if has_bias else None
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    xin = F.interpolate(xr, scale_factor=2, mode='nearest') if ups else xr
    yr = F.conv2d(xin, wr, br, s, p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if has_bias else None
    y = ops.conv2d(xd, wd, bd, s, p, upsample=bool(ups))
    assert y.shape == yr.shape
    y.backward(gy.to(dev))
    tol = 2e-5     # fp32 MFMA = k-ordered fmaf chains; differences are summation-order only
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    if has_bias:
        assert rel_err(bd.grad, br.grad) < tol


# (case, activation): the epilogue activation on the single-pass MFMA path, the split-K combine and the VALU head
@pytest.mark.parametrize("case", [(2, 3, 32, 32, 64, 3, 1, 1, 0, True), (1, 512, 8, 8, 512, 3, 1, 1, 0, True),
                                  (2, 64, 64, 64, 3, 3, 1, 1, 0, True), (2, 64, 32, 32, 3, 3, 1, 1, 0, True)])
@pytest.mark.parametrize("act", ['lrelu', 'tanh', 'lrelu+pool'])
def test_conv2d_fused_activation(dev, case, act):
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, ups, has_bias = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, s, p)
    pool = act.endswith('+pool')
    yr = torch.tanh(pre) if act == 'tanh' else F.leaky_relu(pre, 0.2)
    if pool:
        yr = F.avg_pool2d(yr, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True); bd = b.to(dev).requires_grad_(True)
    kind = ops.ACT_TANH if act == 'tanh' else ops.ACT_LRELU
    y = ops.conv2d(xd, wd, bd, s, p, act_kind=kind, slope=0.2, pool=pool)
    assert y.shape == yr.shape
    y.backward(gy.to(dev))
    tol = 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    assert rel_err(bd.grad, br.grad) < tol


# 4x4 / stride 2 / pad 1 shapes the LDS-patch kernels of csrc/conv_tpatch.h take (plan cfg 6): 2-D tile blocks, whole-sample
# and multi-sample tiles, several 64-row tiles, split-K over channel groups, bias / activation epilogue
TPATCH_CASES = [
    (2, 64, 32, 32, 64, True, 'none'),       # critic block: 4 tiles of 4 x 16 outputs per sample
    (4, 128, 8, 8, 192, True, 'lrelu'),      # 4 x 4 outputs per sample: one tile = 4 samples; 3 row tiles; input gradient too
    (1, 128, 32, 128, 64, False, 'none'),    # 16 x 64 outputs: tiles of 2 x 32; input gradient with 2 row tiles
    (2, 512, 16, 16, 128, True, 'none'),     # deep reduction: split-K over channel groups
    (3, 72, 16, 16, 64, False, 'tanh'),      # whole-sample tiles, 9 channel groups of 8
]


@pytest.mark.parametrize("case", TPATCH_CASES)
def test_conv4x4s2_patch_kernels(dev, case):
    ops = _ops()
    from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD
    B, Cin, H, W, Cout, has_bias, act = case
    assert ops.get_plan((B, Cin, H, W, Cout, 4, 4, 2, 1, 0), dev).mode(CONV_FWD).info.cfg == 6, "not on the 4x4 patch kernel"
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) if has_bias else None
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_bias else None
    yr = F.conv2d(xr, wr, br, 2, 1)
    yr = torch.tanh(yr) if act == 'tanh' else F.leaky_relu(yr, 0.2) if act == 'lrelu' else yr
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if has_bias else None
    kind = {'none': ops.ACT_NONE, 'lrelu': ops.ACT_LRELU, 'tanh': ops.ACT_TANH}[act]
    y = ops.conv2d(xd, wd, bd, 2, 1, act_kind=kind, slope=0.2)
    y.backward(gy.to(dev))
    tol = 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    if has_bias:
        assert rel_err(bd.grad, br.grad) < tol


WPATCH_CASES = [
    (3, 128, 7, 128, 64, 0),      # two 64-channel column blocks, 2 column strips, 7 rows in chunks of 4 + 3
    (2, 64, 16, 64, 128, 0),      # two 64-row blocks
    (2, 64, 6, 64, 64, 1),        # behind the fused nearest upsample: 12 output rows x 128 columns from 6 x 64 stored
    (2, 128, 8, 32, 64, 1),       # upsampled to 16 x 64: one strip, two 64-channel column blocks
]


@pytest.mark.parametrize("case", WPATCH_CASES)
def test_conv3x3_row_ring_weight_gradient(dev, case):
    """the LDS row-ring WGRAD kernel (csrc/conv_wpatch.h) against torch fp64, through the module-level op"""
    ops = _ops()
    from semantic_pyramid_for_image_generation_amd._lib import CONV_WGRAD
    B, Cin, H, W, Cout, ups = case
    plan = ops.get_plan((B, Cin, H, W, Cout, 3, 3, 1, 1, ups), dev)
    assert plan.mode(CONV_WGRAD).info.cfg == 7, "not on the row-ring kernel"
    g = torch.Generator().manual_seed(37)
    x = torch.randn(B, Cin, H, W, generator=g)
    gy = torch.randn(B, Cout, H << ups, W << ups, generator=g)
    xr = F.interpolate(x.double(), scale_factor=2, mode='nearest') if ups else x.double()
    gwr = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), gy.double(), stride=1, padding=1)
    gw = ops._raw_conv_wgrad(plan, gy.to(dev), x.to(dev))
    assert rel_err(gw, gwr) < 2e-5
    gw2 = ops._raw_conv_wgrad(plan, gy.to(dev), x.to(dev))
    assert torch.equal(gw, gw2)                     # fixed summation order: bit-identical from launch to launch
    acc = gw.clone()
    ops._raw_conv_wgrad(plan, gy.to(dev), x.to(dev), into=acc)          # `into`: the combine pass adds to what is there
    assert rel_err(acc, 2 * gwr) < 2e-5


def test_conv_transpose4x4s2_patch_kernel(dev):
    """ConvTranspose2d(4, 2, 1) forward = the per-phase input-gradient kernel (128 output channels: two row tiles); its
    backward = the forward patch kernel."""
    ops = _ops()
    from semantic_pyramid_for_image_generation_amd._lib import CONV_DGRAD
    B, Cin, H, W, Cout = 2, 64, 8, 8, 128
    assert ops.get_plan((B, Cout, 2 * H, 2 * W, Cin, 4, 4, 2, 1, 0), dev).mode(CONV_DGRAD).info.cfg == 6
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    y = ops.conv_transpose2d(xd, wd, None, 2, 1)
    y.backward(gy.to(dev))
    tol = 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol


CONVT_CASES = [
    (2, 128, 1, 1, 512, 4, 1, 0, False),   # G conv_block1: 1x1 -> 4x4
    (2, 64, 4, 4, 64, 4, 2, 1, False),     # G deconv 4x4 s2
    (2, 64, 16, 16, 3, 4, 2, 1, True),     # G default last_conv (bias)
    (1, 6, 5, 7, 10, 4, 2, 1, True),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d_fwd_bwd(dev, case):
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, has_bias = case
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) if has_bias else None
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    yr = F.conv_transpose2d(xr, wr, br, s, p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if has_bias else None
    y = ops.conv_transpose2d(xd, wd, bd, s, p)
    assert y.shape == yr.shape
    y.backward(gy.to(dev))
    tol = 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    if has_bias:
        assert rel_err(bd.grad, br.grad) < tol


def test_conv_double_backward(dev):
    """d/d(w, x) of || d conv(x,w) / dx ||-style penalty: the conv part of the WGAN-GP double backward."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 12, 12, generator=g); w = torch.randn(16, 8, 4, 4, generator=g) * 0.2
    b = torch.randn(16, generator=g)

    import contextlib

    def run(conv, x, w, b, inner=contextlib.nullcontext):
        x = x.clone().requires_grad_(True); w = w.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
        y = conv(x, w, b)
        pred = (y * y).sum() * 0.5
        with inner():       # the HIP path is told that only d pred / d x is wanted (as gradient_penalty does)
            gx, = torch.autograd.grad(pred, x, create_graph=True)
        pen = (gx * gx).sum()
        pen.backward()
        return gx, x.grad, w.grad
    ref = run(lambda x, w, b: F.conv2d(x, w, b, 2, 1), x, w, b)
    out = run(lambda x, w, b: ops.conv2d(x, w, b, 2, 1), x.to(dev), w.to(dev), b.to(dev), ops.input_grads_only)
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 5e-5
    # the same through the fused conv + LeakyReLU epilogue (critic stem), without and with the fused AvgPool2d(2)
    ref = run(lambda x, w, b: F.leaky_relu(F.conv2d(x, w, b, 2, 1), 0.2), x, w, b)
    out = run(lambda x, w, b: ops.conv2d(x, w, b, 2, 1, act_kind=ops.ACT_LRELU, slope=0.2), x.to(dev), w.to(dev), b.to(dev),
              ops.input_grads_only)
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 5e-5
    ref = run(lambda x, w, b: F.avg_pool2d(F.leaky_relu(F.conv2d(x, w, b, 2, 1), 0.2), 2), x, w, b)
    out = run(lambda x, w, b: ops.conv2d(x, w, b, 2, 1, act_kind=ops.ACT_LRELU, slope=0.2, pool=True),
              x.to(dev), w.to(dev), b.to(dev), ops.input_grads_only)
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 5e-5


@pytest.mark.parametrize("shape", [(2, 8, 256), (1, 34, 256), (3, 2, 512)])
@pytest.mark.parametrize("side", [False, True])
def test_critic_stem_kernels(dev, shape, side):
    """Conv2d(3, 64, 3, 1, 1) -> LeakyReLU(0.2) -> AvgPool2d(2) on 256-pixel rows (discriminator.py:105-109): the dedicated
    forward kernel (pooled output + sign bits), the weight/bias-gradient kernel and the un-pooled input-gradient kernel that
    work from (pooled gradient, bits), in a plain backward and in the penalty-style double backward; `side`: parameter
    gradients on the side stream straight into .grad, twice (assign, then accumulate in place)."""
    import contextlib
    ops = _ops()
    B, H, W = shape
    g = torch.Generator().manual_seed(5 + H)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g) * 0.1
    gy = torch.randn(B, 64, H // 2, W // 2, generator=g)

    def ref_f(x, w, b):
        return F.avg_pool2d(F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.2), 2)

    def hip_f(x, w, b):
        return ops.conv2d(x, w, b, 1, 1, act_kind=ops.ACT_LRELU, slope=0.2, pool=True)

    def plain(f, x, w, b, gy, reps, ctx=contextlib.nullcontext):
        x = x.clone().requires_grad_(True); w = w.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
        with ctx():
            for _ in range(reps):
                y = f(x, w, b)
                y.backward(gy)
        return y, x.grad, w.grad, b.grad
    reps = 2 if side else 1
    ref = plain(ref_f, x, w, b, gy, reps)
    out = plain(hip_f, x.to(dev), w.to(dev), b.to(dev), gy.to(dev), reps, ops.async_param_grads if side else contextlib.nullcontext)
    torch.cuda.synchronize()
    for name, o, r in zip(('y', 'gx', 'gw', 'gb'), out, ref):
        assert rel_err(o, r) < 3e-5, (name, rel_err(o, r))

    def penalty(f, x, w, b, inner=contextlib.nullcontext, ctx=contextlib.nullcontext):
        x = x.clone().requires_grad_(True); w = w.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
        y = f(x, w, b)
        pred = (y * y).sum() * 0.5
        with inner():
            gx, = torch.autograd.grad(pred, x, create_graph=True)
        pen = (gx * gx).sum()
        with ctx():
            pen.backward()
        return gx, x.grad, w.grad
    ref = penalty(ref_f, x, w, b)
    out = penalty(hip_f, x.to(dev), w.to(dev), b.to(dev), ops.input_grads_only, ops.async_param_grads if side else contextlib.nullcontext)
    torch.cuda.synchronize()
    for name, o, r in zip(('gx', 'ggx', 'ggw'), out, ref):
        assert rel_err(o, r) < 1e-4, (name, rel_err(o, r))


def test_stem_wgrad_entry_point(dev):
    """sempyr_stem_wgrad on a shape the autograd route does not send it (two column tiles, more tiles than workgroups is
    covered at full size): against the generic un-pool + weight-gradient + channel-sum kernels, with and without accumulate."""
    ops = _ops()
    B, H, W = 3, 4, 512
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 3, H, W, generator=g).to(dev)
    w = (torch.randn(64, 3, 3, 3, generator=g) * 0.2).to(dev)
    b = (torch.randn(64, generator=g) * 0.1).to(dev)
    gp = torch.randn(B, 64, H // 2, W // 2, generator=g).to(dev)
    plan = ops.get_plan((B, 3, H, W, 64, 3, 3, 1, 1, 0), x.device)
    bits = torch.empty(B * 64 * H * W // 32, dtype=torch.int32, device=dev)
    pooled = torch.empty_like(gp)
    ops._raw_conv_fwd(plan, x, w, b, ops.ACT_LRELU, 0.2, pooled, bits=bits, want_y=False)
    gz = ops._pool2_up_raw(gp, 0.25, None, ops.ACT_LRELU, 0.2, bits=bits)
    gw_ref = ops._raw_conv_wgrad(plan, gz, x)
    gb_ref = gz.sum(dim=(0, 2, 3))
    gw, gb = ops._raw_stem_wgrad(plan, gp, bits, 0.2, x, want_b=True)
    assert rel_err(gw, gw_ref) < 2e-5 and rel_err(gb, gb_ref) < 2e-5
    gw2, gb2 = ops._raw_stem_wgrad(plan, gp, bits, 0.2, x, gw.clone(), True, gb.clone())
    assert torch.equal(gw2, gw + gw) and torch.equal(gb2, gb + gb)
    gw3, none = ops._raw_stem_wgrad(plan, gp, bits, 0.2, x)
    assert none is None and torch.equal(gw3, gw)


# (B, Cin, H, W, Cout, ups): 3x3 s1 p1 bias-free convolutions in front of a training-mode BatchNorm (generator.py:41-43)
STATS_CASES = [
    (2, 64, 32, 32, 64, 0),      # 64x64 tiles would be too few: whichever configuration the plan picks
    (4, 64, 64, 64, 64, 0),      # 64 x 128 / 64 x 256 tiles: 64-pixel groups
    (2, 128, 16, 16, 64, 1),     # fused nearest x2 upsample
    (2, 64, 16, 16, 256, 0),
    (16, 64, 32, 32, 128, 0),
]


@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_epilogue_batchnorm_statistics(dev, case):
    """sempyr_conv2d_fwd_stats + sempyr_bn_stats_from_tiles against conv -> BatchNorm2d(train) -> ReLU in torch: output,
    running statistics, and every gradient; and bit-identical y to the plain convolution."""
    ops = _ops()
    B, Cin, H, W, Cout, ups = case
    g = torch.Generator().manual_seed(21 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    w[0] += 0.02          # a channel whose mean is far from zero relative to its spread
    gam = torch.rand(Cout, generator=g) + 0.5; bet = torch.randn(Cout, generator=g) * 0.1
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    rm_r = torch.zeros(Cout); rv_r = torch.ones(Cout)
    xin = F.interpolate(xr, scale_factor=2, mode='nearest') if ups else xr
    pre = F.conv2d(xin, wr, None, 1, 1)
    # (no activation here: a ReLU element at round-off distance from zero may fall on the other side in either
    # implementation and the max-norm gradient comparison below would measure that, not the kernels; the ReLU epilogue of the
    # apply pass is compared forward-only at the end)
    yr = F.batch_norm(pre, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    gd = gam.to(dev).requires_grad_(True); bd = bet.to(dev).requires_grad_(True)
    rm = torch.zeros(Cout, device=dev); rv = torch.ones(Cout, device=dev); nbt = torch.zeros((), dtype=torch.int64, device=dev)
    plan = ops.get_plan((B, Cin, H, W, Cout, 3, 3, 1, 1, ups), xd.device)
    granule = ops.conv_stats_granule(plan)
    pre_d = ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups), stats=True)
    if granule:
        ts, gr_ = ops.tile_stats_of(pre_d)
        assert gr_ == granule and ts.numel() == 2 * B * Cout * (pre_d.shape[2] * pre_d.shape[3] // granule)
        # the partial sums themselves: group sums and squared deviations, exactly what they claim to be
        v = pre_d.detach().double().reshape(B * Cout, -1, granule).cpu()
        tsd = ts.reshape(B * Cout, -1, 2).double().cpu()
        assert float((tsd[..., 0] - v.sum(-1)).abs().max()) < 1e-3
        m2 = ((v - v.mean(-1, keepdim=True)) ** 2).sum(-1)
        assert float(((tsd[..., 1] - m2).abs() / (m2.abs() + 1e-3)).max()) < 1e-4
    else:
        assert ops.tile_stats_of(pre_d) is None
    with torch.no_grad():
        plain = ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups))
    assert torch.equal(plain, pre_d.detach())
    y = ops.norm_act(pre_d, gd, bd, ops.NORM_BATCH, ops.ACT_NONE, 0.0, 1e-5, (rm, rv, nbt, 0.1))
    y.backward(gy.to(dev))
    tol = 3e-5
    assert rel_err(y, yr) < tol
    assert rel_err(rm, rm_r) < tol and rel_err(rv, rv_r) < tol and int(nbt) == 1
    assert rel_err(xd.grad, xr.grad) < 1e-4
    assert rel_err(wd.grad, wr.grad) < 1e-4
    assert rel_err(gd.grad, gr.grad) < 1e-4 and rel_err(bd.grad, br.grad) < 1e-4
    with torch.no_grad():
        y2 = ops.norm_act(pre_d.detach(), gd, bd, ops.NORM_BATCH, ops.ACT_RELU, 0.0, 1e-5, None) if not granule else \
            ops._apply(ops._NormAct, pre_d.detach(), gd.detach(), bd.detach(), ops.NORM_BATCH, ops.ACT_RELU, 0.0, 1e-5, None,
                       ops.tile_stats_of(pre_d))
    assert rel_err(y2, F.relu(yr)) < tol


def test_conv_epilogue_statistics_cover_the_generator(dev):
    """The shapes the statistics epilogue is for: every 3x3 convolution of the res generator at 64x64 and above reports a
    granule (so the training loop does not fall back to the separate statistics pass unnoticed)."""
    ops = _ops()
    d = torch.device(dev)
    for key in [(16, 64, 128, 128, 64, 3, 3, 1, 1, 1), (16, 64, 256, 256, 64, 3, 3, 1, 1, 0), (16, 64, 64, 64, 64, 3, 3, 1, 1, 1),
                (16, 64, 128, 128, 64, 3, 3, 1, 1, 0), (16, 128, 32, 32, 64, 3, 3, 1, 1, 1), (16, 64, 64, 64, 64, 3, 3, 1, 1, 0)]:
        assert ops.conv_stats_granule(ops.get_plan(key, d)) in (32, 64), key


def test_bn_zero_input_add(dev):
    """sempyr_bn_zero_input_add against the three passes it replaces: BatchNorm2d(train) of an all-zero map + ReLU, added to
    another tensor -- output and running buffers bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    for shape in [(3, 8, 6, 10), (2, 64, 16, 16)]:
        a = torch.randn(*shape, generator=g).to(dev)
        C = shape[1]
        gam = (torch.rand(C, generator=g) + 0.5).to(dev); bet = torch.randn(C, generator=g).to(dev)
        rm1 = torch.randn(C, generator=g).to(dev); rv1 = (torch.rand(C, generator=g) + 0.5).to(dev)
        n1 = torch.tensor(3, dtype=torch.int64, device=dev)
        rm2, rv2, n2 = rm1.clone(), rv1.clone(), n1.clone()
        with torch.no_grad():
            z = torch.zeros(shape, device=dev)
            ref = ops.add_act(a, ops.norm_act(z, gam, bet, ops.NORM_BATCH, ops.ACT_RELU, 0.0, 1e-5, (rm1, rv1, n1, 0.1)))
            out = ops.bn_zero_input_add(a, bet, ops.ACT_RELU, 0.0, (rm2, rv2, n2, 0.1))
        assert torch.equal(out, ref)
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2) and int(n2) == 4


NORM_KINDS = ['batch', 'instance', 'layer']


def _torch_norm(kind, x, gam, bet, eps=1e-5):
    if kind == 'batch':
        return F.batch_norm(x, None, None, gam, bet, True, 0.1, eps)
    if kind == 'instance':
        return F.instance_norm(x, None, None, gam, bet, True, 0.1, eps)
    return F.group_norm(x, 1, gam, bet, eps)


@pytest.mark.parametrize("kind", NORM_KINDS)
@pytest.mark.parametrize("shape", [(3, 8, 4, 4), (2, 5, 7, 9), (2, 4, 64, 64), (2, 4, 16, 16), (2, 3, 32, 32), (1, 2, 64, 32), (3, 6, 8, 8), (5, 30, 4, 4)])
@pytest.mark.parametrize("act", ['none', 'lrelu'])
def test_norm_fwd_bwd_bwdbwd(dev, kind, shape, act):
    ops = _ops()
    KIND = {'batch': ops.NORM_BATCH, 'instance': ops.NORM_INSTANCE, 'layer': ops.NORM_LAYER}[kind]
    ACT = ops.ACT_LRELU if act == 'lrelu' else ops.ACT_NONE
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g) * 1.5 + 0.3
    gam = torch.rand(shape[1], generator=g) + 0.5; bet = torch.randn(shape[1], generator=g) * 0.1
    gy0 = torch.randn(shape, generator=g); u0 = torch.randn(shape, generator=g)

    def run(fn, x, gam, bet, gy, u):
        x = x.clone().requires_grad_(True); gam = gam.clone().requires_grad_(True); bet = bet.clone().requires_grad_(True)
        gy = gy.clone().requires_grad_(True)
        y = fn(x, gam, bet)
        gx, ggam, gbet = torch.autograd.grad(y, [x, gam, bet], gy, create_graph=True)
        L = (gx * u).sum()
        r_gy, r_x, r_gam = torch.autograd.grad(L, [gy, x, gam])
        return y, gx, ggam, gbet, r_gy, r_x, r_gam

    def ref_fn(x, gam, bet):
        y = _torch_norm(kind, x, gam, bet)
        return F.leaky_relu(y, 0.2) if act == 'lrelu' else y
    ref = run(ref_fn, x.double(), gam.double(), bet.double(), gy0.double(), u0.double())
    out = run(lambda x, gam, bet: ops.norm_act(x, gam, bet, KIND, ACT, 0.2), x.to(dev), gam.to(dev), bet.to(dev),
              gy0.to(dev), u0.to(dev))
    names = ['y', 'gx', 'ggamma', 'gbeta', 'dd_gy', 'dd_x', 'dd_gamma']
    for n, o, r in zip(names, out, ref):
        assert rel_err(o, r) < 2e-4, n     # fp32 statistics vs an fp64 reference


@pytest.mark.parametrize("kind", NORM_KINDS)
def test_norm_add_norm_act(dev, kind):
    """relu(norm(a) + norm(b)): the fused residual-block tail against the composed torch ops, first-order gradients."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shape = (3, 8, 12, 12)
    a = torch.randn(shape, generator=g) * 2 + 0.5; b = torch.randn(shape, generator=g) - 1.0
    par = [torch.randn(8, generator=g) * 0.5 + 1 for _ in range(2)] + [torch.randn(8, generator=g) * 0.3 for _ in range(2)]
    gy = torch.randn(shape, generator=g)

    def run(fn, dev_):
        t = [v.clone().to(dev_).requires_grad_(True) for v in (a, b, *par)]
        y = fn(*t)
        y.backward(gy.to(dev_))
        return [y] + [v.grad for v in t]
    ref = run(lambda a, b, ga, gb, ba, bb: F.relu(_torch_norm(kind, a, ga, ba) + _torch_norm(kind, b, gb, bb)), 'cpu')
    code = {'batch': ops.NORM_BATCH, 'instance': ops.NORM_INSTANCE, 'layer': ops.NORM_LAYER}[kind]
    out = run(lambda a, b, ga, gb, ba, bb: ops.norm_add_norm_act(a, ga, ba, b, gb, bb, code, ops.ACT_RELU), dev)
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 5e-5


@pytest.mark.parametrize("shape", [(3, 8, 6, 10), (2, 5, 3, 7)])
def test_norm_add_norm_act_half_resolution_branch(dev, shape):
    """relu(BN(a) + BN(up2(b))) with b given BEFORE the nearest upsample: output, gradients and the running statistics
    (whose unbiased variance counts the upsampled element count) against the composed torch ops on the upsampled map."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    N, C, h, w = shape
    a = torch.randn(N, C, 2 * h, 2 * w, generator=g) * 2 + 0.5
    b = torch.randn(N, C, h, w, generator=g) - 1.0
    par = [torch.randn(C, generator=g) * 0.5 + 1 for _ in range(2)] + [torch.randn(C, generator=g) * 0.3 for _ in range(2)]
    gy = torch.randn(a.shape, generator=g)
    rm_r, rv_r = torch.zeros(C), torch.ones(C)
    tr = [v.clone().requires_grad_(True) for v in (a, b, *par)]
    bu = F.interpolate(tr[1], scale_factor=2, mode='nearest')
    yr = F.relu(F.batch_norm(tr[0], None, None, tr[2], tr[4], True, 0.1, 1e-5) + F.batch_norm(bu, rm_r, rv_r, tr[3], tr[5], True, 0.1, 1e-5))
    yr.backward(gy)
    td = [v.clone().to(dev).requires_grad_(True) for v in (a, b, *par)]
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    y = ops.norm_add_norm_act(td[0], td[2], td[4], td[1], td[3], td[5], ops.NORM_BATCH, ops.ACT_RELU,
                              state_b=(rm, rv, nbt, 0.1), b_half=True)
    y.backward(gy.to(dev))
    assert rel_err(y, yr) < 5e-5
    for o, r in zip(td, tr):
        assert rel_err(o.grad, r.grad) < 5e-5
    assert rel_err(rm, rm_r) < 1e-5 and rel_err(rv, rv_r) < 1e-5 and int(nbt) == 1


def test_batchnorm_running_stats(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 6, 5, 5, generator=g) * 2 + 1
    bn = torch.nn.BatchNorm2d(6)
    bn.train()
    yr = bn(x)
    rm = torch.zeros(6, device=dev); rv = torch.ones(6, device=dev); nbt = torch.zeros((), dtype=torch.long, device=dev)
    y = ops.norm_act(x.to(dev), bn.weight.detach().to(dev), bn.bias.detach().to(dev), ops.NORM_BATCH, ops.ACT_NONE, 0.0,
                     1e-5, (rm, rv, nbt, 0.1))
    assert rel_err(y, yr) < 1e-5
    assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5
    assert int(nbt) == 1


def test_bn_eval_act_residual(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 6, 8, 8, generator=g); res = torch.randn(2, 6, 8, 8, generator=g)
    gam, bet = torch.rand(6, generator=g) + 0.5, torch.randn(6, generator=g)
    rm, rv = torch.randn(6, generator=g), torch.rand(6, generator=g) + 0.5
    xr = x.clone().requires_grad_(True); rr = res.clone().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rm, rv, gam, bet, False, 0.1, 1e-5) + rr)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True); rd = res.to(dev).requires_grad_(True)
    y = ops.bn_eval_act(xd, gam.to(dev), bet.to(dev), rm.to(dev), rv.to(dev), 1e-5, res=rd, act_kind=ops.ACT_RELU)
    y.backward(gy.to(dev))
    assert rel_err(y, yr) < 1e-5 and rel_err(xd.grad, xr.grad) < 1e-5 and rel_err(rd.grad, rr.grad) < 1e-5


# conv -> eval BatchNorm -> (+ residual) -> activation as one launch: the single-pass MFMA kernels (generic gather, 3x3 patch
# with and without the fused upsample), the split-K combine, a ragged M tile and the 7x7 stem
@pytest.mark.parametrize("case", [(2, 64, 32, 32, 64, 3, 1, 1, 0), (2, 64, 16, 16, 64, 3, 1, 1, 1), (2, 64, 16, 16, 128, 3, 2, 1, 0),
                                  (2, 64, 16, 16, 128, 1, 2, 0, 0), (1, 512, 8, 8, 512, 3, 1, 1, 0), (1, 5, 9, 11, 7, 3, 1, 1, 0),
                                  (2, 3, 64, 64, 64, 7, 2, 3, 0), (2, 64, 16, 16, 64, 1, 1, 0, 1)])
@pytest.mark.parametrize("with_res,act", [(True, 'relu'), (False, 'relu'), (False, 'none'), ('half', 'relu')])
def test_conv_bn_eval_fused(dev, case, with_res, act):
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, ups = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    gam, bet = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    rm, rv = torch.randn(Cout, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.5
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode='nearest') if ups else xr
    pre = F.batch_norm(F.conv2d(xin, wr, None, s, p), rm, rv, gam, bet, False, 0.1, 1e-5)
    half = with_res == 'half'           # the residual is given at half resolution and read through a nearest upsample
    if half and (pre.shape[2] % 2 or pre.shape[3] % 2):
        pytest.skip("half-resolution residual needs even output height and width")
    rshape = (pre.shape[0], pre.shape[1], pre.shape[2] // 2, pre.shape[3] // 2) if half else pre.shape
    res = torch.randn(rshape, generator=g) if with_res else None
    rr = res.clone().requires_grad_(True) if with_res else None
    if with_res:
        pre = pre + (F.interpolate(rr, scale_factor=2, mode='nearest') if half else rr)
    yr = F.relu(pre) if act == 'relu' else pre
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    rd = res.to(dev).requires_grad_(True) if with_res else None
    y = ops.conv_bn_eval_act(xd, wd, gam.to(dev), bet.to(dev), rm.to(dev), rv.to(dev), 1e-5, s, p, upsample=bool(ups), res=rd,
                             act_kind=ops.ACT_RELU if act == 'relu' else ops.ACT_NONE, res_half=half)
    assert y.shape == yr.shape
    y.backward(gy.to(dev))
    ops.join_side_streams()
    tol = 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    if with_res:
        assert rel_err(rd.grad, rr.grad) < tol
    # the same call without autograd (eval_model.py runs under no_grad)
    with torch.no_grad():
        y2 = ops.conv_bn_eval_act(xd, wd, gam.to(dev), bet.to(dev), rm.to(dev), rv.to(dev), 1e-5, s, p, upsample=bool(ups),
                                  res=rd, act_kind=ops.ACT_RELU if act == 'relu' else ops.ACT_NONE, res_half=half)
    assert torch.equal(y2, y)


def test_conv_bn_eval_rejects_small_heads(dev):
    ops = _ops()
    from semantic_pyramid_for_image_generation_amd._lib import SempyrError
    x = torch.randn(1, 8, 16, 16, device=dev); w = torch.randn(3, 8, 3, 3, device=dev)
    one = torch.ones(3, device=dev)
    with pytest.raises(SempyrError):
        ops.conv_bn_eval_act(x, w, one, one, one, one, 1e-5, 1, 1)


@pytest.mark.parametrize("case", [(2, 64, 16, 16, 128, 4, 2, 1, 0), (1, 5, 9, 11, 7, 3, 1, 1, 0), (2, 64, 64, 64, 64, 3, 1, 1, 0)])
def test_wgrad_and_bias_grad_accumulate_in_place(dev, case):
    """`accumulate` of sempyr_conv2d_wgrad / sempyr_channel_sum: gw += wgrad(gy, x) equals a separate add, bit for bit
    (few- and many-slab combine kernels)."""
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, ups = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    plan = ops.get_plan((B, Cin, H, W, Cout, k, k, s, p, ups), dev)
    gy = torch.randn(B, Cout, plan.Ho, plan.Wo, generator=g).to(dev)
    old = torch.randn(Cout, Cin, k, k, generator=g).to(dev)
    fresh = ops._raw_conv_wgrad(plan, gy, x)
    acc = old.clone()
    out = ops._raw_conv_wgrad(plan, gy, x, into=acc)
    assert out is acc and torch.equal(acc, old + fresh)
    oldb = torch.randn(Cout, generator=g).to(dev)
    accb = oldb.clone()
    ops._channel_sum_raw(gy, into=accb)
    assert torch.equal(accb, oldb + ops._channel_sum_raw(gy))


def test_elementwise_and_pools(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    a = torch.randn(2, 5, 6, 10, generator=g); b = torch.randn(2, 5, 6, 10, generator=g)
    ad = a.to(dev); bd = b.to(dev)
    assert torch.equal(ops.mul(ad, bd).cpu(), a * b)
    assert rel_err(ops.add_act(ad, bd, ops.ACT_RELU), F.relu(a + b)) == 0
    assert rel_err(ops.act(ad, ops.ACT_LRELU, 0.2), F.leaky_relu(a, 0.2)) < 1e-7
    assert rel_err(ops.act(ad, ops.ACT_TANH), torch.tanh(a)) < 1e-6
    assert rel_err(ops.avg_pool2(ad), F.avg_pool2d(a, 2)) < 1e-6
    assert torch.equal(ops.upsample2(ad).cpu(), F.interpolate(a, scale_factor=2, mode='nearest'))
    # backward of pools / act (incl. double backward of lrelu and avg-pool, both linear in the incoming grad)
    ar = a.clone().requires_grad_(True); adr = a.to(dev).requires_grad_(True)
    gy = torch.randn(2, 5, 3, 5, generator=g)
    F.avg_pool2d(F.leaky_relu(ar, 0.2), 2).backward(gy)
    ops.avg_pool2(ops.act(adr, ops.ACT_LRELU, 0.2)).backward(gy.to(dev))
    assert rel_err(adr.grad, ar.grad) < 1e-6
    # max pool 3/2/1 incl. odd sizes
    for shape in [(2, 3, 8, 8), (1, 2, 7, 9)]:
        x = torch.randn(shape, generator=g)
        xr = x.clone().requires_grad_(True); xd = x.to(dev).requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1); y = ops.max_pool3s2(xd)
        assert torch.equal(y.cpu(), yr)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy); y.backward(gy.to(dev))
        assert rel_err(xd.grad, xr.grad) < 1e-6
    # normalize (true division) + its backward
    mean = torch.tensor([0.485, 0.456, 0.406]); std = torch.tensor([0.229, 0.224, 0.225])
    x = torch.rand(2, 3, 6, 6, generator=g)
    xr = x.clone().requires_grad_(True); xd = x.to(dev).requires_grad_(True)
    yr = (xr - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    y = ops.normalize(xd, mean.to(dev), std.to(dev))
    assert rel_err(y, yr) < 1e-7
    gy = torch.randn(yr.shape, generator=g); yr.backward(gy); y.backward(gy.to(dev))
    assert rel_err(xd.grad, xr.grad) < 1e-6
    # lerp
    al = torch.rand(2, 1, 1, 1, generator=g)
    assert rel_err(ops.lerp_rows(ad, bd, al.to(dev)), a + al * (b - a)) < 1e-6
    # global average pool
    assert rel_err(ops.global_avg_pool(ad), F.adaptive_avg_pool2d(a, 1)) < 1e-6


def test_reductions(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    for n in [(2, 1, 1, 1), (3, 7, 5, 5), (2, 64, 64, 64)]:
        a = torch.randn(n, generator=g); b = torch.randn(n, generator=g); m = (torch.rand(n, generator=g) > 0.4).float()
        ar = a.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        adr = a.to(dev).requires_grad_(True); bdr = b.to(dev).requires_grad_(True)
        lr = -ar.mean() * 1.0 + F.l1_loss(ar * m, br * m) * 3 + F.l1_loss(ar, br)
        ld = ops.mean(adr, -1.0) + ops.l1_mean(adr, bdr, m.to(dev)) * 3 + ops.l1_mean(adr, bdr)
        assert rel_err(ld, lr) < 1e-5
        lr.backward(); ld.backward()
        assert rel_err(adr.grad, ar.grad) < 1e-5 and rel_err(bdr.grad, br.grad) < 1e-5
    # gradient-penalty reduction and its backward
    x = torch.randn(3, 3, 16, 16, generator=g)
    xr = x.clone().requires_grad_(True); xd = x.to(dev).requires_grad_(True)
    gr = ((xr.view(3, -1).norm(p=2, dim=1) - 1) ** 2).mean(); gd = ops.gp_loss(xd)
    assert rel_err(gd, gr) < 1e-5
    (gr * 10).backward(); (gd * 10).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-5


def test_no_cpu_fallback():
    from semantic_pyramid_for_image_generation_amd import ops, _lib
    with pytest.raises(_lib.SempyrError):
        ops.conv2d(torch.randn(1, 3, 8, 8), torch.randn(4, 3, 3, 3), None, 1, 1)


def test_multi_tensor_repack_matches_single_packs(dev):
    """sempyr_conv2d_pack_weights_many (every cached layout of a module in one launch, behind its optimizer step) fills the
    cached buffers with exactly what the per-tensor pack kernel writes: tables with and without padding rows / columns"""
    ops = _ops()
    from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD, CONV_DGRAD
    g = torch.Generator().manual_seed(41)
    shapes = [(2, 64, 16, 16, 128, 4, 4, 2, 1, 0),    # critic block: 4 stride phases in the input-gradient table
              (2, 128, 8, 8, 64, 3, 3, 1, 1, 1),      # generator conv behind the fused upsample
              (2, 3, 32, 32, 64, 7, 7, 2, 3, 0),      # classifier stem: K = 147, padded tables
              (2, 72, 8, 8, 40, 3, 3, 1, 1, 0),       # Cout = 40: padded columns
              (2, 512, 4, 4, 1, 4, 4, 1, 0, 0)]       # critic head: one output row
    params = []
    for key in shapes:
        plan = ops.get_plan(key, dev)
        w = torch.nn.Parameter(torch.randn(key[4], key[1], key[5], key[6], generator=g).to(dev))
        for mode in (CONV_FWD, CONV_DGRAD):
            ops._packed_weights(plan.noups() if mode == CONV_DGRAD else plan, w, mode)
        params.append((plan, w))
    with torch.no_grad():
        for _, w in params:
            w.mul_(-1.5).add_(0.25)
    ops.repack_cached([w for _, w in params], None)
    torch.cuda.synchronize()
    n = 0
    for plan, w in params:
        for wkey, hit in w._sempyr_pack.items():
            pl, mode = hit[5], hit[6]
            fresh = torch.full_like(hit[1], float('nan'))
            ops.check(ops.L.sempyr_conv2d_pack_weights(pl.desc_ref, mode, w.data_ptr(), pl.mode(mode).wsrc.data_ptr(), fresh.data_ptr(),
                                                       ops._stream()), "sempyr_conv2d_pack_weights")
            assert torch.equal(hit[1], fresh), (pl.key, mode)
            n += 1
    assert n == 2 * len(shapes)
