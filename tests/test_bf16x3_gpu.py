"""The opt-in split-bf16 convolution arithmetic (sempyr_conv_desc.math = SEMPYR_MATH_BF16X3, ops.CONV_MATH / SEMPYR_CONV_MATH=bf16x3):
3x3 stride-1 convolutions and their input gradients form every product as hi*hi + hi*lo + lo*hi of bf16 halves on
v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The headline path stays exact fp32; this variant has its own stated
tolerances, checked here against the same oracles:
  * one convolution (forward, input gradient) against fp64: <= 2e-5 of max-abs (measured 4e-6 .. 6e-6; the fp32 kernels 4e-7),
    and NOT bit-equal to the fp32 kernel's result (the variant really ran);
  * generator forward, critic step and generator step of the headline configuration against the fp64 oracle: images and losses
    within the north star's 1e-3;
  * parameter gradients of those steps: within the measured-noise bound of tests/test_steps_gpu.py (3x the distance of the fp32
    CPU runs from fp64) widened by BF16X3_GRAD_FACTOR -- the split keeps 16 significant bits per operand, not 24."""
import copy

import pytest
import torch
import torch.nn.functional as F

import test_steps_gpu as T
from oracle import nets, recipes

pytestmark = pytest.mark.gpu

# how much wider than the fp32 path's bound (3x the fp32-vs-fp64 noise) the split-bf16 gradients are allowed to be.  Measured
# (printed by the tests, -s): the worst tensor of the critic step sits at 1.5x the fp32 path's bound, of the generator step
# (diversity loss on) at 2.6x.
BF16X3_GRAD_FACTOR = 3.0


class _math:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from semantic_pyramid_for_image_generation_amd import ops
        self.ops, self.saved = ops, ops.CONV_MATH
        ops.CONV_MATH = self.mode

    def __exit__(self, *exc):
        self.ops.CONV_MATH = self.saved


@pytest.mark.parametrize("case", [((2, 64, 64, 64), 64, 0), ((2, 64, 32, 32), 64, 1), ((2, 128, 32, 32), 64, 0),
                                  ((2, 512, 8, 8), 256, 1), ((3, 32, 16, 64), 64, 0)])
def test_conv3x3_split_bf16_vs_fp64(dev, case):
    from semantic_pyramid_for_image_generation_amd import ops
    shape, cout, ups = case
    B, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g); w = torch.randn(cout, C, 3, 3, generator=g) * 0.05
    gy = torch.randn(B, cout, H << ups, W << ups, generator=g)

    def run(mode):
        with _math(mode):
            xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
            y = ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups))
            gx, gw = torch.autograd.grad(y, [xd, wd], gy.to(dev))
        return y.detach(), gx, gw
    y16, gx16, gw16 = run(1)
    y32, gx32, gw32 = run(0)
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    xi = F.interpolate(x64, scale_factor=2, mode='nearest') if ups else x64
    yr = F.conv2d(xi, w64, None, 1, 1)
    gxr, gwr = torch.autograd.grad(yr, [x64, w64], gy.double())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert rel(y16, yr.detach()) <= 2e-5 and rel(gx16, gxr) <= 2e-5 and rel(gw16, gwr) <= 2e-6
    assert rel(y32, yr.detach()) <= 2e-6
    if C % 16 == 0 and cout % 64 == 0 and cout % 128 != 0:
        assert not torch.equal(y16, y32)          # the split-bf16 kernel took the forward launch
    assert torch.equal(gw16, gw32)                # weight gradients stay on the fp32 kernels


def test_generator_images_and_steps_within_1e3(dev):
    """BASELINE configuration 2 (res generator, instance_norm critic) at batch 2 with the split-bf16 convolutions: generated
    images max-abs <= 1e-3 against the oracle, critic-step and generator-step losses <= 1e-3, gradients against the widened
    measured-noise bound."""
    with _math(1):
        tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, 'res', 'instance_norm', False, 71)
        # images
        with torch.no_grad():
            want = tr.G(T._d(z), T._d(feats), T._d(masks), training=True)
            got = G(z.to(dev), [f.to(dev) for f in feats], [m.to(dev) for m in masks])
        err = float((got.double().cpu() - want).abs().max())
        print(f"[bf16x3] generated images: max-abs error {err:.2e}")
        assert err <= 1e-3
        # the train-mode forward above moved the BatchNorm buffers on both sides alike; steps from here
        d_loss_fn, g_loss_fn = tm.get_wgan_losses_fn()
        crit = tm.L1Loss()
        real = (images - 0.5) / 0.5
        refs32 = T._fp32_runs(tr, lambda t: t.d_step(T.j(real), T.j(feats), masks, T.j(z), alpha))
        d64_before = {k: v.detach().clone() for k, v in tr.d.items()}
        want = tr.d_step(T._d(real), T._d(feats), T._d(masks), T._d(z), T._d(alpha))
        d_before = copy.deepcopy(D.state_dict())
        with T._CaptureLine(tm) as cap:
            got = tm.train_discriminator(G, D, d_loss_fn, T._Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)), real.to(dev),
                                         [f.to(dev) for f in feats], [m.to(dev) for m in masks], z=z.to(dev), alpha=alpha.to(dev))
        for k in want:
            assert abs(float(got[k].detach()) - float(want[k])) <= 1e-3 * max(1.0, abs(float(want[k]))), k
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in D.named_parameters()}
        # a LeakyReLU pre-activation at round-off distance from zero that lands on the other side: as in test_steps_gpu (the fake
        # batch differs from the fp32 path's by ~1e-5 here, so WHICH elements do is a draw of its own)
        flips, dist = T._lrelu_flips(D, d_before, d64_before, 'instance_norm', 'res', real, cap, dev)
        print(f"[bf16x3-d] LeakyReLU pre-activations that changed side: {flips} (largest {dist:.1e} of the tensor maximum)")
        assert dist < 1e-4, f"{flips} LeakyReLU signs differ from fp64, one at {dist:.1e} of its tensor's maximum: not round-off"
        saved = dict(T.ALLOW)
        try:
            T.ALLOW[('bf16x3-d', '')] = ('factor', BF16X3_GRAD_FACTOR)
            T._grads_close(grads, tr.d, [r.d for r in refs32], tr.d_params, 'bf16x3-d', widen=10.0 if flips else 1.0)
        finally:
            T.ALLOW.clear(); T.ALLOW.update(saved)


def test_generator_step_gradients(dev):
    with _math(1):
        tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, 'res', 'instance_norm', True, 72)
        _, g_loss_fn = tm.get_wgan_losses_fn()
        crit = tm.L1Loss()
        refs32 = T._fp32_runs(tr, lambda t: t.g_step(T.j(feats), masks, 3, T.j(z), T.j(z2)))
        want = tr.g_step(T._d(feats), T._d(masks), 3, T._d(z), T._d(z2))
        got = tm.train_generator(G, D, g_loss_fn, T._Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999)), 2, [f.to(dev) for f in feats],
                                 crit, 3, C, tm.Normalize(recipes.CLF_MEAN, recipes.CLF_STD), crit, crit, [m.to(dev) for m in masks],
                                 [2], z=z.to(dev), z2=z2.to(dev))
        for k in want:
            assert abs(float(got[k].detach()) - float(want[k])) <= 1e-3 * max(1.0, abs(float(want[k]))), k
        saved = dict(T.ALLOW)
        try:
            T.ALLOW[('bf16x3-g', '')] = ('factor', BF16X3_GRAD_FACTOR)
            T._grads_close({k: p.grad for k, p in G.named_parameters()}, tr.g, [r.g for r in refs32], tr.g_params, 'bf16x3-g')
        finally:
            T.ALLOW.clear(); T.ALLOW.update(saved)
