"""WGAN-GP training of the semantic-pyramid generator — drop-in mirror of the reference
`train_model.py`: same function names/signatures (losses, gradient penalty, train steps, mask
setters), the same module-global `args`, CLI flags, output directories and checkpoint files; the
tensor work runs on the HIP operators (`ops.py`).

Deliberate, result-neutral deviations from the reference: gradients that the reference
computes and never uses (classifier weights; discriminator weights during the generator step) are
not computed.
"""
import argparse
import math
import os
import random
import time

import torch
import torch.nn as nn
from . import optim

from . import ops, parallel
from .classifier import Classifier  # noqa: F401
from .discriminator import Discriminator
from .generator import Generator

# branches of a step run on two streams (train_discriminator / train_generator): the parameter-gradient accumulation
# nodes then see gradients from both, which is intended
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

args = None     # set by the CLI (or by the caller): needs .use_diversity_loss and .gradient_penalty_weight


def print_gpu_details():
    print('Is GPU available:', torch.cuda.is_available())
    if torch.cuda.is_available():
        print('Num of GPUs:', torch.cuda.device_count())


################################################################
# WGAN-GP loss functions and helpers (reference train_model.py:35-76)
################################################################

def get_wgan_losses_fn():
    def d_loss_fn(r_logit, f_logit):
        return ops.mean(r_logit, -1.0), ops.mean(f_logit, 1.0)

    def g_loss_fn(f_logit):
        return ops.mean(f_logit, -1.0)

    d_loss_fn._sempyr_wgan = True
    return d_loss_fn, g_loss_fn


def sample_line(real, fake, alpha=None):
    """real + alpha*(fake-real), alpha ~ U[0,1) per sample drawn on real's device (reference :51-55).
    `alpha` may be injected (parity tests)."""
    if alpha is None:
        shape = [real.size(0)] + [1] * (real.dim() - 1)
        alpha = torch.rand(shape, device=real.device)
    return ops.lerp_rows(real, fake, alpha)


def _row_scale(x, coef):
    """out[r, :] = x[r, :] * coef[r]"""
    rows = x.size(0)
    c = torch.cat([coef.reshape(-1).float(), torch.zeros(rows, dtype=torch.float32, device=x.device)])
    out = torch.empty_like(x)
    ops.check(ops.L.sempyr_affine_act(x.data_ptr(), 0, c.data_ptr(), c.data_ptr() + 4 * rows, 0, out.data_ptr(), rows,
                                      x.numel() // rows, ops.ACT_NONE, 0.0, ops._stream()), "sempyr_affine_act")
    return out


def _row_dot(a, b):
    """[rows] vector of <a[r, :], b[r, :]>"""
    rows = a.size(0)
    prod = torch.empty_like(a)
    ops.check(ops.L.sempyr_mul(a.data_ptr(), b.data_ptr(), prod.data_ptr(), a.numel(), ops._stream()), "sempyr_mul")
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    scratch = torch.empty(rows, dtype=torch.float32, device=a.device)
    ops.check(ops.L.sempyr_channel_sum(prod.data_ptr(), 1, rows, a.numel() // rows, 1.0, out.data_ptr(), 0, scratch.data_ptr(),
                                       ops._stream()), "sempyr_channel_sum")
    return out


class _RowNorm(torch.autograd.Function):
    """norms[r] = ||x[r, :]||_2"""

    @staticmethod
    def forward(ctx, x):
        rows = x.size(0)
        x = ops._req(x, "norm input")
        norms = torch.empty(rows, dtype=torch.float32, device=x.device)
        gp = torch.empty((), dtype=torch.float32, device=x.device)
        ops.check(ops.L.sempyr_gp_fwd(x.data_ptr(), rows, x.numel() // rows, norms.data_ptr(), gp.data_ptr(), 0, ops._stream()),
                  "sempyr_gp_fwd")
        ctx.save_for_backward(x, norms)
        return norms

    @staticmethod
    def backward(ctx, g):
        x, norms = ctx.saved_tensors
        return _RowNormBwd.apply(g.contiguous(), x, norms)


class _RowNormBwd(torch.autograd.Function):
    """gx[r, :] = g[r] * x[r, :] / n[r] (0 where n[r] == 0, as torch's norm backward); differentiable in g and x, which is
    what a gradient penalty composed by hand from norm() (reference train_model.py:58-66 under create_graph=True) needs."""

    @staticmethod
    def forward(ctx, g, x, norms):
        inv = torch.where(norms > 0, 1.0 / norms, torch.zeros_like(norms))
        ctx.save_for_backward(g, x, inv)
        return _row_scale(x, g * inv)

    @staticmethod
    def backward(ctx, u):
        g, x, inv = ctx.saved_tensors
        u = u.contiguous()
        xu = _row_dot(x, u)                                   # <x_r, u_r>
        g_g = xu * inv if ctx.needs_input_grad[0] else None
        g_x = None
        if ctx.needs_input_grad[1]:
            # d/dx [g x / n] . u = g (u / n - x <x,u> / n^3)
            a = _row_scale(u, g * inv)
            b = _row_scale(x, -(g * xu) * inv * inv * inv)
            g_x = torch.empty_like(x)
            ops.check(ops.L.sempyr_axpby(a.data_ptr(), b.data_ptr(), 1.0, 1.0, g_x.data_ptr(), x.numel(), ops._stream()),
                      "sempyr_axpby")
        return g_g, g_x, None


def norm(x):
    """per-sample L2 norm over all non-batch dims (reference :58-60); twice differentiable."""
    return _RowNorm.apply(x)


def one_mean_gp(grad):
    """mean((||grad||-1)^2) (reference :63-66) as one fused reduction."""
    return ops.gp_loss(grad)


def gradient_penalty(f, real, fake, alpha=None):
    """WGAN-GP penalty (reference :69-76): d f(x)/dx at x on the real-fake line, with create_graph=True.
    Only the input gradient is wanted, so parameter gradients are skipped in this inner pass."""
    x = sample_line(real, fake, alpha).detach()
    x.requires_grad = True
    ops.graph_only_input_grad(x)      # d loss / d x is never read: the plain backward pass stops at the critic's first convolution
    pred = f(x)
    with ops.input_grads_only():
        grad = torch.autograd.grad(pred, x, grad_outputs=ops.full_like(pred, 1.0), create_graph=True)[0]
    return one_mean_gp(grad)


class L1Loss(nn.Module):
    """nn.L1Loss() stand-in on the HIP reduction; `masked(a, b, m)` = L1(a*m, b*m) in one pass."""

    def forward(self, a, b):
        return ops.l1_mean(a, b)

    def masked(self, a, b, m):
        return ops.l1_mean(a, b, m)


class Normalize:
    """transforms.Normalize(mean, std) for (B,C,H,W) batches (reference :240-245)."""

    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)
        self._dev = {}

    def __call__(self, x):
        t = self._dev.get(x.device)
        if t is None:
            t = (torch.tensor(self.mean, dtype=torch.float32, device=x.device),
                 torch.tensor(self.std, dtype=torch.float32, device=x.device))
            self._dev[x.device] = t
        return ops.normalize(x, t[0], t[1])


################################################################
# Training and sampling (reference train_model.py:83-165)
################################################################

def _ensure_train(module):
    """module.train() as the reference calls it at the top of every step (train_model.py:86-87,123-124), skipped when
    the module is already in training mode: the recursive flag walk over ~100 submodules cost 7 ms of host time per
    cycle.  (eval() / train() are only ever applied to whole modules on this path.)"""
    if not module.training:
        module.train()


class _frozen:
    """Temporarily stop tracking parameter gradients of a module (grads the reference computes but never uses)."""

    def __init__(self, *modules):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


DEAD_TAPS = os.environ.get("SEMPYR_DEAD_TAPS", "1") != "0"


def live_taps(masks):
    """Deepest classifier tap (1..4) whose mask is not known to be all-zero; 0 when every mask is.  The reference multiplies
    the taps at and above the drawn layer by all-zero masks (quirk Q2: train_model.py:197-209) both where the generator reads
    them (generator.py:126-129) and in the feature loss (train_model.py:212-222), so nothing deeper than this level can
    change any result: the classifier need not compute it (Classifier.forward(..., taps_upto)).  None: shortcut off, or a
    mask carries no tag (masks built by other means): compute everything."""
    if not DEAD_TAPS:
        return None
    live = 0
    for i in range(1, 5):
        m = masks[i]
        if m is None or not hasattr(m, '_sempyr_all_zero'):
            return None
        if not m._sempyr_all_zero:
            live = i
    return live


def _device_of(features, masks):
    for t in list(features) + list(masks):
        if t is not None:
            return t.device
    raise ValueError("no tensor to take the device from")


def train_generator(generator, discriminator, generator_loss_fn, generator_optimizer, batch_size, features, criterion_features,
                    features_to_train, classifier, normalizer_clf, criterion_diversity_n, criterion_diversity_d, masks,
                    train_type, epsilon=10e-4, z=None, z2=None, fake_images=None):
    """One generator step (reference :83-119).  `z` / `z2` may be injected (parity tests); by default they are
    drawn on the device generator as in the reference.  `fake_images`: the step's own forward `generator(z, features,
    masks)` WITH its autograd graph, computed ahead by the caller (`fake_for_generator` inside a `prefetch` block: that
    forward does not depend on the critic step in front of it); `z` must then be the noise it was made from."""
    device = _device_of(features, masks)
    _ensure_train(generator)
    _ensure_train(discriminator)
    if fake_images is not None and z is None:
        raise ValueError("train_generator: fake_images needs the z it was generated from")
    if z is None:
        z = torch.randn(batch_size, 128, 1, 1, device=device)

    with _frozen(discriminator, classifier):
        if fake_images is None:
            fake_images = generator(z, features, masks)
        two_stream = ops.TWO_STREAM_GENERATOR and fake_images.is_cuda
        if two_stream:
            # the critic branch and the classifier-feature branch only share fake_images: two streams (forward and, through
            # the autograd engine, backward), joined before the losses are summed
            main = torch.cuda.current_stream(device)
            side = ops.side_stream(device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                generator_loss = generator_loss_fn(discriminator(fake_images))
            fake_images.record_stream(side)
        else:
            generator_loss = generator_loss_fn(discriminator(fake_images))

        upto = live_taps(masks) if train_type != 1 else None
        if upto == 0:
            # every level of the feature loss is masked to zero: the loss is exactly 0 with an exactly-zero gradient
            features_loss = ops.full_like(generator_loss, 0.0)
        else:
            fake_images_clf = normalizer_clf(fake_images)
            _, fake_features = classifier(fake_images_clf) if upto is None else classifier(fake_images_clf, taps_upto=upto)
            features_loss = getLossByTrainType(features, masks, train_type, features_to_train, fake_features, criterion_features)
        if two_stream:
            main.wait_stream(side)
            generator_loss.record_stream(main)

        total_loss = ops.lincomb([generator_loss, features_loss])
        loses_dictionary = {'g_loss': generator_loss, 'features_loss': features_loss, 'total_loss': total_loss}

        if args.use_diversity_loss:
            if z2 is None:
                z2 = torch.randn(batch_size, 128, 1, 1, device=device)
            fake_images2 = generator(z2, features, masks)
            diversity_loss = criterion_diversity_n(z, z2) / (criterion_diversity_d(fake_images, fake_images2) + epsilon)
            total_loss = ops.lincomb([total_loss, diversity_loss])
            loses_dictionary['diversity loss'] = diversity_loss
            loses_dictionary['total loss'] = total_loss

        parallel.zero_gradients(generator)
        with ops.async_param_grads():
            total_loss.backward(_one(device))
    parallel.sync_gradients(generator)
    generator_optimizer.step()
    mark_generator_updated(generator, device)
    return loses_dictionary


_GEN_UPDATED = {}        # id(generator) -> event recorded behind its last optimizer step


def mark_generator_updated(generator, device):
    """Record, on the current stream, that the generator's parameters have just been updated (the prefetch stream waits for
    this before the next generator forward).  Inside a graph capture nothing is recorded -- an event of a capture means
    nothing outside it; whoever replays the capture calls this after the replay (loop.py)."""
    if device.type == 'cuda':
        if torch.cuda.is_current_stream_capturing():
            _GEN_UPDATED.pop(id(generator), None)
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            _GEN_UPDATED[id(generator)] = ev


class prefetch:
    """`with prefetch(generator, device) as pf:` — run the part of a loop iteration that does not depend on the previous
    critic step (classifier taps, masks, and the detached generator forward of train_discriminator, reference
    train_model.py:308-318,125-126) on a separate stream.  That stream waits only for the generator's last update, so the
    work overlaps the previous iteration's critic backward, which the host has already queued.  Tensors created inside are
    handed to the other streams by `pf.hand_over(...)`; `train_discriminator(..., fake_images=...)` then waits for the block.
    Outside CUDA (or with SEMPYR_PREFETCH=0) the block runs in place."""

    def __init__(self, generator, device):
        self.on = ops.PREFETCH and device.type == 'cuda' and not torch.cuda.is_current_stream_capturing()
        self.generator, self.device = generator, device
        self.done = None

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.device)
            self.stream = ops.prefetch_stream(self.device)
            # (the generator's last update is waited for where the generator is first used — fake_for_discriminator /
            # fake_for_generator — so the classifier taps and masks of the iteration after a generator step overlap that
            # step's backward as well)
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
        return self

    def hand_over(self, *tensors):
        """Mark tensors created in the block as used by the library's other streams (caching-allocator bookkeeping)."""
        if self.on:
            for t in tensors:
                for item in (t if isinstance(t, (list, tuple)) else [t]):
                    if item is None:
                        continue
                    for st in self.consumers:
                        item.record_stream(st)

    def __exit__(self, *exc):
        if self.on:
            self.done = torch.cuda.Event()
            self.done.record(self.stream)
            self.ctx.__exit__(*exc)
            self.main.wait_event(self.done)

    @property
    def consumers(self):
        return [self.main] + ops.consumer_streams(self.device)[1:]


def _wait_for_generator_update(generator, device):
    """Order the current stream behind the generator's last optimizer step (it may have been issued on another stream)."""
    if device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
        ev = _GEN_UPDATED.get(id(generator))
        if ev is not None:
            torch.cuda.current_stream(device).wait_event(ev)


def fake_for_generator(generator, features, masks, z):
    """The generator step's own forward (reference train_model.py:93), with its autograd graph, for a caller that runs it
    ahead of the critic step inside a `prefetch` block; pass the result and `z` to train_generator."""
    _ensure_train(generator)
    _wait_for_generator_update(generator, _device_of(features, masks))
    return generator(z, features, masks)


def fake_for_discriminator(generator, batch_size, features, masks, z=None):
    """The detached generator forward of a critic step (reference train_model.py:125-126): z from the CPU generator."""
    device = _device_of(features, masks)
    _ensure_train(generator)
    if z is None:
        z = _cpu_randn_to(device, batch_size, 128, 1, 1)
    _wait_for_generator_update(generator, device)
    with torch.no_grad():
        return generator(z, features, masks)


_PINNED = {}


def _cpu_randn_to(device, *shape, out=None):
    """`torch.randn(*shape).to(device)` (same draws from the CPU generator as the reference, train_model.py:125) without
    the host-device synchronisation of a pageable copy: the noise is drawn into one of a small ring of pinned buffers and
    copied asynchronously; a buffer is reused only after the event recorded behind its last copy has completed.
    `out`: an existing device tensor that receives the noise."""
    if device.type != 'cuda':
        z = torch.randn(*shape).to(device)
        return z if out is None else out.copy_(z)
    ring = _PINNED.setdefault((device.index, shape), {'bufs': [], 'next': 0})
    if len(ring['bufs']) < 8:
        ring['bufs'].append([torch.empty(shape, dtype=torch.float32).pin_memory(), None])
        slot = ring['bufs'][-1]
    else:
        slot = ring['bufs'][ring['next']]
        ring['next'] = (ring['next'] + 1) % len(ring['bufs'])
        slot[1].synchronize()
    torch.randn(shape, out=slot[0])
    z = slot[0].to(device, non_blocking=True) if out is None else out.copy_(slot[0], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    slot[1] = ev
    return z


_ONE = {}


def _one(device):
    """the constant 1.0 that seeds a backward pass (loss.backward() would launch a fill for it every time)"""
    t = _ONE.get(device)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _ONE[device] = t
    return t


def _critic_losses_batched(discriminator_loss_fn, logits, n_real, n_fake):
    """discriminator_loss_fn(logits[:n_real], logits[n_real:]) for the WGAN loss pair of get_wgan_losses_fn, computed from the
    batched logits without slicing; any other loss function gets the slices."""
    if getattr(discriminator_loss_fn, '_sempyr_wgan', False):
        return ops.mean_parts(logits, (n_real, n_fake), (-1.0, 1.0))
    return discriminator_loss_fn(logits[:n_real], logits[n_real:])


def _samplewise(module):
    """True when no submodule computes statistics across the batch (BatchNorm), i.e. f(cat(a, b)) == cat(f(a), f(b))."""
    return not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in module.modules())


def train_discriminator(generator, discriminator, discriminator_loss_fn, discriminator_optimizer, real_images, features, masks,
                        z=None, alpha=None, fake_images=None):
    """One critic step (reference :122-145): real/fake logits + 10 * gradient penalty (double backward).
    `fake_images` may carry the result of fake_for_discriminator() computed ahead of time (see `prefetch`)."""
    device = real_images.device
    _ensure_train(generator)
    _ensure_train(discriminator)
    if fake_images is None:                                               # reference: forward with graph, then .detach()
        fake_images = fake_for_discriminator(generator, real_images.shape[0], features, masks, z)   # CPU-generator z (Q7)

    samplewise = _samplewise(discriminator)

    def logits_loss():
        if samplewise:
            # no layer mixes samples (instance / layer norm or none): one pass over both batches gives the same logits as
            # the reference's two passes and fills the small late layers twice as well
            n, m = real_images.shape[0], fake_images.shape[0]
            logits = discriminator(ops.concat_rows([real_images, fake_images]))
            return _critic_losses_batched(discriminator_loss_fn, logits, n, m)
        return discriminator_loss_fn(discriminator(real_images), discriminator(fake_images))

    if ops.TWO_STREAM_CRITIC and real_images.is_cuda and samplewise:
        # the logit branch and the gradient-penalty branch share nothing but the weights: run them on two streams (the
        # autograd engine replays each branch's backward on the stream of its forward), so their many mid-size kernels
        # fill each other's idle CUs.  Not with a BatchNorm critic: its running statistics are read-modify-written by
        # every pass, in the reference's order real, fake, x_hat
        main = torch.cuda.current_stream(device)
        side = ops.side_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            real_images_d_loss, fake_images_d_loss = logits_loss()
        real_images.record_stream(side); fake_images.record_stream(side)
        gp = gradient_penalty(discriminator, real_images, fake_images, alpha)
        main.wait_stream(side)
        real_images_d_loss.record_stream(main); fake_images_d_loss.record_stream(main)
    else:
        real_images_d_loss, fake_images_d_loss = logits_loss()
        gp = gradient_penalty(discriminator, real_images, fake_images, alpha)

    # (real + fake) + gp * weight, and real + fake for the log: one launch each
    discriminator_loss = ops.lincomb([real_images_d_loss, fake_images_d_loss, gp], [1.0, 1.0, args.gradient_penalty_weight])
    d_loss = ops.lincomb([real_images_d_loss.detach(), fake_images_d_loss.detach()])

    parallel.zero_gradients(discriminator)
    with ops.async_param_grads():
        discriminator_loss.backward(_one(device))
    parallel.sync_gradients(discriminator)
    discriminator_optimizer.step()
    return {'d_loss': d_loss, 'gp': gp}


def sample(generator, z, features, masks):
    """eval-mode generator forward without autograd (reference :148-154)."""
    with torch.no_grad():
        generator.eval()
        return generator(z, features, masks)


def isolate_layer(fixed_features, selected_layer, device):
    """keep the selected entry, zero the others (reference :157-165)."""
    res = [torch.clone(f) for f in fixed_features]
    for i in range(len(fixed_features)):
        if i != selected_layer:
            res[i] = torch.zeros(res[i].shape, device=device)
    return res


################################################################
# Masks (reference train_model.py:172-222) — index arithmetic is bit-exact with the reference
################################################################

def setCroppedMask(random_crop, mask, window_len1=0.45, window_len2=0.45, window_data=0):
    x_start = math.floor((random_crop[0] / 100) * mask.shape[2])
    y_start = math.floor((random_crop[1] / 100) * mask.shape[3])
    x_end = x_start + math.floor(window_len1 * mask.shape[2])
    y_end = y_start + math.floor(window_len2 * mask.shape[3])
    mask[:, :, x_start:x_end, y_start:y_end] = window_data
    return mask


def crop_window(random_crop, H, W, window_len1=0.45, window_len2=0.45):
    """(x0, x1, y0, y1): the rows / columns setCroppedMask overwrites, with the clamping of Python slicing."""
    x_start = math.floor((random_crop[0] / 100) * H)
    y_start = math.floor((random_crop[1] / 100) * W)
    x_end = x_start + math.floor(window_len1 * H)
    y_end = y_start + math.floor(window_len2 * W)
    return min(x_start, H), min(x_end, H), min(y_start, W), min(y_end, W)


def mask_specs(shapes, train_type_is_1, random_layer_idx, random_crop=None):
    """One (base, win, x0, x1, y0, y1) per mask: the mask is `base` except rows [x0,x1) x cols [y0,y1) = `win`.
    setMasksPart1 (train type 1: the chosen layer whole, every other layer blocked) or setMasksPart2 (type 2: layers
    below the chosen one pass with a zeroed window; the chosen layer and the ones above it are blocked -- the reference's
    loop overwrites the ones it first stores at the chosen index, quirk Q2)."""
    specs = []
    for idx, shp in enumerate(shapes):
        if train_type_is_1:
            specs.append((1.0, 1.0, 0, 0, 0, 0) if idx == random_layer_idx else (0.0, 0.0, 0, 0, 0, 0))
        elif idx < random_layer_idx:
            specs.append((1.0, 0.0) + crop_window(random_crop, shp[2], shp[3]))
        else:
            specs.append((0.0, 0.0, 0, 0, 0, 0))
    return specs


def spec_words(specs):
    words = []
    for base, win, x0, x1, y0, y1 in specs:
        words += [ops.float_bits(base), ops.float_bits(win), x0, x1, y0, y1]
    return words


def tag_mask(mask, spec):
    """Host-side note of what a mask holds, for consumers that can skip work on an all-zero mask without reading it.
    Only the all-zero property is recorded (it is fixed by train type and layer; window positions change per iteration)."""
    base, win, x0, x1, y0, y1 = spec
    mask._sempyr_all_zero = (base == 0.0 and (win == 0.0 or x0 >= x1 or y0 >= y1))
    return mask


def _masks_from_specs(masks, device, specs):
    device = torch.device(device)
    if device.type != 'cuda':
        # host path (CPU tensors only appear in host-logic tests and tooling; the train loop runs on the GPU)
        for idx, (base, win, x0, x1, y0, y1) in enumerate(specs):
            m = torch.full(masks[idx].shape, base, dtype=torch.float32, device=device)
            m[:, :, x0:x1, y0:y1] = win
            masks[idx] = tag_mask(m, specs[idx])
        return
    out = [torch.empty(m.shape, dtype=torch.float32, device=device) for m in masks]
    spec_dev = torch.empty(6 * len(out), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        ops.store_words(spec_dev, spec_words(specs))
        ops.build_masks(out, spec_dev)
    for idx, m in enumerate(out):
        masks[idx] = tag_mask(m, specs[idx])


def setMasksPart1(masks, device, random_layer_idx):
    """train type 1: pass the chosen layer whole, block every other layer (reference :184-194); one launch for all masks."""
    _masks_from_specs(masks, device, mask_specs([m.shape for m in masks], True, random_layer_idx))


def setMasksPart2(masks, device, random_layer_idx):
    """train type 2 (reference :197-209): layers below the chosen one pass with a zeroed random window; the chosen layer
    and the ones above it are blocked (quirk Q2).  Draws the crop with two random.randint(20, 40) calls."""
    random_crop = (random.randint(20, 40), random.randint(20, 40))
    _masks_from_specs(masks, device, mask_specs([m.shape for m in masks], False, random_layer_idx, random_crop))


def getLossByTrainType(features, masks, train_type, features_to_train, outputs_images_features, criterion_features):
    """feature-reconstruction loss (reference :212-222).  `train_type` is the list `random.choices` returns, so
    `train_type == 1` is False and the masked multi-layer branch always runs (quirk Q1)."""
    masked = getattr(criterion_features, 'masked', None)

    def term(i):
        if masked is not None:
            return masked(features[i], outputs_images_features[i], masks[i])
        return criterion_features(ops.mul(features[i], masks[i]), ops.mul(outputs_images_features[i], masks[i]))

    if train_type == 1:
        terms = [criterion_features(features[features_to_train], outputs_images_features[features_to_train])]
    else:
        # a level whose mask is all zeros contributes |f*0 - g*0| = 0 exactly: when its taps were not computed
        # (live_taps) the term is left out, which leaves the sum bit for bit what it was (x + 0 = x)
        terms = [term(i) for i in range(1, features_to_train + 1)
                 if not (features[i] is None or outputs_images_features[i] is None)]
        if not terms:
            raise ValueError("getLossByTrainType: no classifier tap was computed for any level of the loss")
    if all(t.is_cuda for t in terms):
        return ops.lincomb(terms, None, 1 / features_to_train, 200.0)        # (sum * (1/L)) * 200 in one launch
    loss_features = terms[0]
    for t in terms[1:]:
        loss_features = loss_features + t
    return loss_features * (1 / features_to_train) * 200


################################################################
# CLI (reference train_model.py:225-410)
################################################################

def load_classifier(path, device):
    from .compat import load_classifier as _load
    return _load(path, device)


def _main():
    from . import imageio
    print_gpu_details()
    device = parallel.init_device()
    rank, world = parallel.rank(), parallel.world_size()

    normalizer_clf = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    normalizer_discriminator = Normalize([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])
    print("set image folder")
    train_set = imageio.ImageFolder(args.train_path, image_size=256, crop_size=256, raw=imageio.DEVICE_TRANSFORM)
    print('set data loader')
    train_loader = imageio.make_loader(train_set, args.batch_size, shuffle=True, num_workers=4, drop_last=True,
                                       rank=rank, world_size=world)

    classifier = load_classifier(args.classifier_path, device)
    classifier.eval()
    for p in classifier.parameters():
        p.requires_grad_(False)
    generator = Generator(gen_type=args.gen_type)
    discriminator = Discriminator(args.discriminator_norm, dis_type=args.gen_type)
    if args.generator_path is not None:
        generator.load_state_dict(torch.load(args.generator_path, map_location='cpu', weights_only=True))
    else:
        generator.init_weights()
    if args.discriminator_path is not None:
        discriminator.load_state_dict(torch.load(args.discriminator_path, map_location='cpu', weights_only=True))
    else:
        discriminator.init_weights()
    generator.to(device)
    discriminator.to(device)
    parallel.broadcast_module(generator)
    parallel.broadcast_module(discriminator)

    dp = {}
    if world > 1:
        # data parallel: .grad of every parameter is a view of one persistent flat buffer, summed in place by ONE RCCL
        # all-reduce per optimizer step; Adam reads the sum as a mean (grad_scale) and leaves zeros behind (zero_grads)
        parallel.flatten_grads(generator); parallel.flatten_grads(discriminator, bucket_bytes=parallel.BUCKET_BYTES)
        dp = dict(grad_scale=1.0 / world, zero_grads=True)
    generator_optimizer = optim.Adam(generator.parameters(), lr=args.lr, betas=(0.5, 0.999), capturable=True, **dp)
    discriminator_optimizer = optim.Adam(discriminator.parameters(), lr=args.lr, betas=(0.5, 0.999), capturable=True, **dp)
    from .loop import TrainLoop
    # the loop body (reference :300-349) lives in TrainLoop: same draws in the same order, the same step functions above,
    # replayed from captured graphs after the first occurrence of each variant
    loop = TrainLoop(generator, discriminator, classifier, generator_optimizer, discriminator_optimizer, args.batch_size, device,
                     discriminator_steps=args.discriminator_steps, train1_prob=args.train1_prob, fixed_layer=args.fixed_layer,
                     normalizer_clf=normalizer_clf, normalizer_discriminator=normalizer_discriminator,
                     use_graphs=False if world > 1 else None)       # (a collective inside a captured body is not rehearsed)
    loop.prepare()          # warm + capture every body variant now: the loop never mixes eager and replayed iterations

    starting_time = time.time()
    iterations = 0
    outputs_dir = os.path.join('wgan-gp_models', args.model_name)
    temp_results_dir = os.path.join(outputs_dir, 'temp_results')
    models_dir = os.path.join(outputs_dir, 'models_checkpoint')
    writer = None
    if rank == 0:
        for d in (outputs_dir, temp_results_dir, models_dir):
            os.makedirs(d, exist_ok=True)
        writer = imageio.SummaryWriter(os.path.join(outputs_dir, 'summaries'))

    z = torch.randn(args.batch_size, 128, 1, 1).to(device)
    z2 = torch.randn(args.batch_size, 128, 1, 1).to(device)
    fixed_features = fixed_masks = fixed_features_diversity = None
    first_iter = True

    def to_loop_input(data):
        images, _ = data
        if isinstance(images, (list, tuple)):
            # raw decoded pixels: Resize / CenterCrop / ToTensor on the GPU (sempyr_resize_crop_normalize), on the loop's
            # prefetch stream, where the iteration's first consumers run
            with torch.cuda.stream(ops.prefetch_stream(device)):
                images = imageio.batch_images(images, device)
        return images

    print("Starting Training Loop...")
    for epoch in range(args.epochs):
        if hasattr(train_loader, 'sampler') and hasattr(train_loader.sampler, 'set_epoch'):
            train_loader.sampler.set_epoch(epoch)          # data-parallel runs: a new permutation every epoch, as the reference's shuffle
        for data, data_next in _with_next(train_loader, to_loop_input, loop.lookahead):
            iterations += 1
            if iterations % 30 == 1:
                print('epoch:', epoch, ', iter', iterations, 'start, time =', time.time() - starting_time, 'seconds')
                starting_time = time.time()
            # (a pinned host batch goes to the loop as it is: the upload is issued on the loop's prefetch stream, in front of
            # the work that reads it, instead of on this stream behind the previous iteration's backward pass.  The loop is
            # also given the NEXT batch: it launches that iteration's stage A ahead of this one's stage B -- never across a
            # generator step, and not in the iterations that end in the sampling / checkpoint blocks below, which must see no
            # work in flight that reads the generator or updates its BatchNorm buffers)
            images = data
            # the next iteration that samples from / checkpoints the networks below: nothing behind it may be in flight then
            to_sampling = (1 - iterations) % 1000
            if data_next is not None:
                data_next = data_next[:to_sampling] or None
            out = loop.iteration(images, log=writer is not None, next_images=data_next)
            if first_iter and rank == 0:
                first_iter = False
                torch.cuda.current_stream(device).wait_stream(ops.prefetch_stream(device))     # (device-transformed images are produced there)
                with torch.no_grad():       # every tap of the first batch (the loop only computes the taps an iteration reads)
                    _, features = classifier(normalizer_clf(images.to(device)))
                images_discriminator = out['images_d']
                fixed_features = [torch.clone(f) for f in features]
                fixed_masks = [torch.ones(f.shape, device=device) for f in features]
                fixed_features_diversity = [torch.clone(f) for f in features]
                for f in fixed_features_diversity:
                    for j in range(f.shape[0]):
                        f[j] = f[j % 8]
                grid = imageio.make_grid(images_discriminator, padding=2, normalize=True, nrow=8)
                imageio.save_image(grid, os.path.join(temp_results_dir, 'original_images.jpg'))
                orig_images_diversity = torch.clone(images_discriminator)
                for i in range(orig_images_diversity.shape[0]):
                    orig_images_diversity[i] = orig_images_diversity[i % 8]
                grid = imageio.make_grid(orig_images_diversity, padding=2, normalize=True, nrow=8)
                imageio.save_image(grid, os.path.join(temp_results_dir, 'original_images_diversity.jpg'))
            if writer is not None and (iterations % 30 == 1):          # scalars leave the device in batches: one sync per 30 iterations
                for it, scalars in loop.pop_logged():
                    for tag, v in scalars.items():
                        writer.add_scalar(tag, v, global_step=it if tag.startswith('D/') else it // 5 + 1)
                        if it == iterations:
                            print('{}: {:.6f}'.format(tag.split('/', 1)[1], v))

            if rank == 0 and iterations % 1000 == 1:
                torch.save(generator.state_dict(), models_dir + '/' + args.model_name + 'G')
                torch.save(discriminator.state_dict(), models_dir + '/' + args.model_name + 'D')
            if rank == 0 and args.keep_temp_results:
                if iterations < 10000 and iterations % 1000 == 1 or iterations % 2000 == 1:
                    fake_images = fake_images_diversity = None
                    for i in range(1, 5):
                        one_layer_mask = isolate_layer(fixed_masks, i, device)
                        if fake_images is None:
                            fake_images = sample(generator, z, fixed_features, one_layer_mask)
                            fake_images_diversity = sample(generator, z, fixed_features_diversity, one_layer_mask)
                        else:
                            fake_images = torch.vstack((fake_images, sample(generator, z, fixed_features, one_layer_mask)))
                            fake_images_diversity = torch.vstack((fake_images_diversity,
                                                                  sample(generator, z2, fixed_features_diversity, one_layer_mask)))
                    generator.train()
                    grid = imageio.make_grid(fake_images, padding=2, normalize=True, nrow=8)
                    imageio.save_image(grid, os.path.join(temp_results_dir, 'res_iter_{}.jpg'.format(iterations // 1000)))
                    grid = imageio.make_grid(fake_images_diversity, padding=2, normalize=True, nrow=8)
                    imageio.save_image(grid, os.path.join(temp_results_dir, 'div_iter_{}.jpg'.format(iterations // 1000)))
                if iterations % 20000 == 1:
                    torch.save(generator.state_dict(), models_dir + '/' + args.model_name + 'G_' + str(iterations // 15000))
                    torch.save(discriminator.state_dict(), models_dir + '/' + args.model_name + 'D_' + str(iterations // 15000))


def _with_next(iterable, prepare, n=1):
    """(prepare(item), [prepare() of the up to n items that follow]) over `iterable`; every item is prepared once"""
    it = iter(iterable)
    window = []
    for item in it:
        window.append(prepare(item))
        if len(window) == n + 1:
            yield window[0], window[1:]
            window.pop(0)
    while window:
        yield window[0], (window[1:] or None)
        window.pop(0)


def build_parser():
    parser = argparse.ArgumentParser('choose hyperparameters')
    parser.add_argument('--batch-size', default=16, type=int)
    parser.add_argument('--lr', default=0.0002, type=float)
    parser.add_argument('--epochs', default=100, type=int)
    parser.add_argument('--model-name')
    parser.add_argument('--discriminator-norm', default='instance_norm', choices=['batch_norm', 'instance_norm', 'layer_norm'])
    parser.add_argument('--gradient-penalty-weight', type=float, default=10.0)
    parser.add_argument('--discriminator-steps', type=int, default=5)
    parser.add_argument('--gen-type', default='res', choices=['default', 'res'])
    parser.add_argument('--train-path', default='/home/dcor/datasets/places365')
    parser.add_argument('--classifier-path', default='/home/dcor/ronmokady/workshop21/team1/try/classifier18')
    parser.add_argument('--generator-path', help='None for random init, if path is given the generator initialize to the given model')
    parser.add_argument('--discriminator-path', help='None for random init, if path is given the discriminator initialize to the given model')
    parser.add_argument('--train1-prob', default=0.6, type=float, help='use value of 1 in order of use only train type 1 (no masks)')
    parser.add_argument('--fixed-layer', type=int, help='train only this layer')
    parser.add_argument('--keep-temp-results', action='store_true')
    parser.add_argument('--use-diversity-loss', action='store_true')
    return parser


def main(argv=None):
    global args
    args = build_parser().parse_args(argv)
    print(args)
    if args.model_name is None:
        print("Must specify model name")
        exit(1)
    out_dir = './wgan-gp_models/' + args.model_name
    if parallel.rank_from_env() == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(out_dir + '/hyperparams.txt', 'w') as file:
            for key, value in vars(args).items():
                file.write(key + ': {}\n'.format(value))
    _main()


if __name__ == '__main__':
    main()
