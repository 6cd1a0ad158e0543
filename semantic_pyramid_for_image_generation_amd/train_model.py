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

    return d_loss_fn, g_loss_fn


def sample_line(real, fake, alpha=None):
    """real + alpha*(fake-real), alpha ~ U[0,1) per sample drawn on real's device (reference :51-55).
    `alpha` may be injected (parity tests)."""
    if alpha is None:
        shape = [real.size(0)] + [1] * (real.dim() - 1)
        alpha = torch.rand(shape, device=real.device)
    return ops.lerp_rows(real, fake, alpha)


class _RowNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        rows = x.size(0)
        x = x.contiguous()
        norms = torch.empty(rows, dtype=torch.float32, device=x.device)
        gp = torch.empty((), dtype=torch.float32, device=x.device)
        ops.check(ops.L.sempyr_gp_fwd(x.data_ptr(), rows, x.numel() // rows, norms.data_ptr(), gp.data_ptr(), ops._stream()),
                  "sempyr_gp_fwd")
        ctx.save_for_backward(x, norms)
        return norms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, norms = ctx.saved_tensors
        rows = x.size(0)
        coef = torch.cat([g / norms, torch.zeros_like(norms)])
        out = torch.empty_like(x)
        ops.check(ops.L.sempyr_affine_act(x.data_ptr(), 0, coef.data_ptr(), coef.data_ptr() + 4 * rows, 0, out.data_ptr(), rows,
                                          x.numel() // rows, ops.ACT_NONE, 0.0, ops._stream()), "sempyr_affine_act")
        return out


def norm(x):
    """per-sample L2 norm over all non-batch dims (reference :58-60)."""
    return _RowNorm.apply(x)


def one_mean_gp(grad):
    """mean((||grad||-1)^2) (reference :63-66) as one fused reduction."""
    return ops.gp_loss(grad)


def gradient_penalty(f, real, fake, alpha=None):
    """WGAN-GP penalty (reference :69-76): d f(x)/dx at x on the real-fake line, with create_graph=True.
    Only the input gradient is wanted, so parameter gradients are skipped in this inner pass."""
    x = sample_line(real, fake, alpha).detach()
    x.requires_grad = True
    pred = f(x)
    with ops.input_grads_only():
        grad = torch.autograd.grad(pred, x, grad_outputs=torch.ones_like(pred), create_graph=True)[0]
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


def train_generator(generator, discriminator, generator_loss_fn, generator_optimizer, batch_size, features, criterion_features,
                    features_to_train, classifier, normalizer_clf, criterion_diversity_n, criterion_diversity_d, masks,
                    train_type, epsilon=10e-4, z=None, z2=None, fake_images=None):
    """One generator step (reference :83-119).  `z` / `z2` may be injected (parity tests); by default they are
    drawn on the device generator as in the reference.  `fake_images`: the step's own forward `generator(z, features,
    masks)` WITH its autograd graph, computed ahead by the caller (`fake_for_generator` inside a `prefetch` block: that
    forward does not depend on the critic step in front of it); `z` must then be the noise it was made from."""
    device = features[1].device
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

        fake_images_clf = normalizer_clf(fake_images)
        _, fake_features = classifier(fake_images_clf)
        features_loss = getLossByTrainType(features, masks, train_type, features_to_train, fake_features, criterion_features)
        if two_stream:
            main.wait_stream(side)
            generator_loss.record_stream(main)

        total_loss = generator_loss + features_loss
        loses_dictionary = {'g_loss': generator_loss, 'features_loss': features_loss, 'total_loss': total_loss}

        if args.use_diversity_loss:
            if z2 is None:
                z2 = torch.randn(batch_size, 128, 1, 1, device=device)
            fake_images2 = generator(z2, features, masks)
            diversity_loss = criterion_diversity_n(z, z2) / (criterion_diversity_d(fake_images, fake_images2) + epsilon)
            total_loss = total_loss + diversity_loss
            loses_dictionary['diversity loss'] = diversity_loss
            loses_dictionary['total loss'] = total_loss

        generator.zero_grad()
        with ops.async_param_grads():
            total_loss.backward()
    parallel.sync_gradients(generator)
    generator_optimizer.step()
    if device.type == 'cuda':
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        _GEN_UPDATED[id(generator)] = ev
    return loses_dictionary


_GEN_UPDATED = {}        # id(generator) -> event recorded behind its last optimizer step


class prefetch:
    """`with prefetch(generator, device) as pf:` — run the part of a loop iteration that does not depend on the previous
    critic step (classifier taps, masks, and the detached generator forward of train_discriminator, reference
    train_model.py:308-318,125-126) on a separate stream.  That stream waits only for the generator's last update, so the
    work overlaps the previous iteration's critic backward, which the host has already queued.  Tensors created inside are
    handed to the other streams by `pf.hand_over(...)`; `train_discriminator(..., fake_images=...)` then waits for the block.
    Outside CUDA (or with SEMPYR_PREFETCH=0) the block runs in place."""

    def __init__(self, generator, device):
        self.on = ops.PREFETCH and device.type == 'cuda'
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
    if device.type == 'cuda':
        ev = _GEN_UPDATED.get(id(generator))
        if ev is not None:
            torch.cuda.current_stream(device).wait_event(ev)


def fake_for_generator(generator, features, masks, z):
    """The generator step's own forward (reference train_model.py:93), with its autograd graph, for a caller that runs it
    ahead of the critic step inside a `prefetch` block; pass the result and `z` to train_generator."""
    _ensure_train(generator)
    _wait_for_generator_update(generator, features[1].device)
    return generator(z, features, masks)


def fake_for_discriminator(generator, batch_size, features, masks, z=None):
    """The detached generator forward of a critic step (reference train_model.py:125-126): z from the CPU generator."""
    device = features[1].device
    _ensure_train(generator)
    if z is None:
        z = _cpu_randn_to(device, batch_size, 128, 1, 1)
    _wait_for_generator_update(generator, device)
    with torch.no_grad():
        return generator(z, features, masks)


_PINNED = {}


def _cpu_randn_to(device, *shape):
    """`torch.randn(*shape).to(device)` (same draws from the CPU generator as the reference, train_model.py:125) without
    the host-device synchronisation of a pageable copy: the noise is drawn into one of a small ring of pinned buffers and
    copied asynchronously; a buffer is reused only after the event recorded behind its last copy has completed."""
    if device.type != 'cuda':
        return torch.randn(*shape).to(device)
    ring = _PINNED.setdefault((device.index, shape), {'bufs': [], 'next': 0})
    if len(ring['bufs']) < 8:
        ring['bufs'].append([torch.empty(shape, dtype=torch.float32).pin_memory(), None])
        slot = ring['bufs'][-1]
    else:
        slot = ring['bufs'][ring['next']]
        ring['next'] = (ring['next'] + 1) % len(ring['bufs'])
        slot[1].synchronize()
    torch.randn(shape, out=slot[0])
    z = slot[0].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    slot[1] = ev
    return z


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

    def logits_loss():
        if _samplewise(discriminator):
            # no layer mixes samples (instance / layer norm or none): one pass over both batches gives the same logits as
            # the reference's two passes and fills the small late layers twice as well
            n = real_images.shape[0]
            logits = discriminator(torch.cat([real_images, fake_images]))
            return discriminator_loss_fn(logits[:n], logits[n:])
        return discriminator_loss_fn(discriminator(real_images), discriminator(fake_images))

    if ops.TWO_STREAM_CRITIC and real_images.is_cuda:
        # the logit branch and the gradient-penalty branch share nothing but the weights: run them on two streams (the
        # autograd engine replays each branch's backward on the stream of its forward), so their many mid-size kernels
        # fill each other's idle CUs
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

    discriminator_loss = (real_images_d_loss + fake_images_d_loss) + gp * args.gradient_penalty_weight

    discriminator.zero_grad()
    with ops.async_param_grads():
        discriminator_loss.backward()
    parallel.sync_gradients(discriminator)
    discriminator_optimizer.step()
    return {'d_loss': real_images_d_loss + fake_images_d_loss, 'gp': gp}


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


def setMasksPart1(masks, device, random_layer_idx):
    """train type 1: pass the chosen layer whole, block every other layer."""
    for idx, mask in enumerate(masks):
        masks[idx] = torch.ones(mask.shape, device=device) if idx == random_layer_idx else mask * 0


def setMasksPart2(masks, device, random_layer_idx):
    """train type 2: layers below the chosen one pass with a zeroed random window; the chosen layer and the ones above
    it are blocked (the reference's loop overwrites the ones it first stores at the chosen index — quirk Q2)."""
    random_crop = (random.randint(20, 40), random.randint(20, 40))
    masks[random_layer_idx] = torch.ones(masks[random_layer_idx].shape, device=device)
    for idx, mask in enumerate(masks):
        if idx < random_layer_idx:
            masks[idx] = setCroppedMask(random_crop, torch.ones(mask.shape, device=device))
        else:
            masks[idx] = mask * 0


def getLossByTrainType(features, masks, train_type, features_to_train, outputs_images_features, criterion_features):
    """feature-reconstruction loss (reference :212-222).  `train_type` is the list `random.choices` returns, so
    `train_type == 1` is False and the masked multi-layer branch always runs (quirk Q1)."""
    masked = getattr(criterion_features, 'masked', None)

    def term(i):
        if masked is not None:
            return masked(features[i], outputs_images_features[i], masks[i])
        return criterion_features(ops.mul(features[i], masks[i]), ops.mul(outputs_images_features[i], masks[i]))

    if train_type == 1:
        loss_features = criterion_features(features[features_to_train], outputs_images_features[features_to_train])
    else:
        loss_features = term(1)
        for i in range(2, features_to_train + 1):
            loss_features = loss_features + term(i)
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
    train_set = imageio.ImageFolder(args.train_path, image_size=256, crop_size=256)
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

    criterion_discriminator, criterion_generator = get_wgan_losses_fn()
    criterion_features = L1Loss()
    criterion_diversity_n = L1Loss()
    criterion_diversity_d = L1Loss()
    generator_optimizer = optim.Adam(generator.parameters(), lr=args.lr, betas=(0.5, 0.999))
    discriminator_optimizer = optim.Adam(discriminator.parameters(), lr=args.lr, betas=(0.5, 0.999))

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
    pending = []                 # (tag, step, device scalar) logged without a per-iteration host sync
    print("Starting Training Loop...")
    for epoch in range(args.epochs):
        for data in train_loader:
            train_type = random.choices([1, 2], [args.train1_prob, 1 - args.train1_prob])
            iterations += 1
            if iterations % 30 == 1:
                print('epoch:', epoch, ', iter', iterations, 'start, time =', time.time() - starting_time, 'seconds')
                starting_time = time.time()
            images, _ = data
            # everything up to the critic's own forward is independent of the previous critic step: it runs on the
            # prefetch stream and overlaps that step's backward (see `prefetch`)
            pf = prefetch(generator, device)
            pf.__enter__()
            images = images.to(device, non_blocking=True)
            images_discriminator = normalizer_discriminator(images)
            images_clf = normalizer_clf(images)
            with torch.no_grad():
                _, features = classifier(images_clf)
            if first_iter and rank == 0:
                first_iter = False
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
            features_to_train = random.randint(1, len(features) - 2) if args.fixed_layer is None else args.fixed_layer
            masks = [features[i].clone() for i in range(len(features))]
            if train_type == 1:
                setMasksPart1(masks, device, features_to_train)
            else:
                setMasksPart2(masks, device, features_to_train)
            fake_images = fake_for_discriminator(generator, images.shape[0], features, masks)
            pf.hand_over(images, images_discriminator, features, masks, fake_images)
            z_g = fake_g = None
            if iterations % args.discriminator_steps == 1 and pf.on:
                # this iteration ends with a generator step: its forward does not depend on the critic step either
                z_g = torch.randn(images.shape[0], 128, 1, 1, device=device)
                fake_g = fake_for_generator(generator, features, masks, z_g)
                pf.hand_over(z_g, fake_g)
            pf.__exit__(None, None, None)
            discriminator_loss_dict = train_discriminator(generator, discriminator, criterion_discriminator,
                                                          discriminator_optimizer, images_discriminator, features, masks,
                                                          fake_images=fake_images)
            for k, v in discriminator_loss_dict.items():
                pending.append(('D/%s' % k, iterations, v.detach()))
                if iterations % 30 == 1:
                    print('{}: {:.6f}'.format(k, v))
            if iterations % args.discriminator_steps == 1:
                generator_loss_dict = train_generator(generator, discriminator, criterion_generator, generator_optimizer,
                                                      images.shape[0], features, criterion_features, features_to_train, classifier,
                                                      normalizer_clf, criterion_diversity_n, criterion_diversity_d, masks, train_type,
                                                      z=z_g, fake_images=fake_g)
                for k, v in generator_loss_dict.items():
                    pending.append(('G/%s' % k, iterations // 5 + 1, v.detach()))
                    if iterations % 30 == 1:
                        print('{}: {:.6f}'.format(k, v))
            if writer is not None and (iterations % 30 == 1):          # flush scalars in batches: one sync per 30 iterations
                for tag, step, v in pending:
                    writer.add_scalar(tag, float(v), global_step=step)
                pending = []
            elif writer is None:
                pending = []

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
                    grid = imageio.make_grid(fake_images, padding=2, normalize=True, nrow=8)
                    imageio.save_image(grid, os.path.join(temp_results_dir, 'res_iter_{}.jpg'.format(iterations // 1000)))
                    grid = imageio.make_grid(fake_images_diversity, padding=2, normalize=True, nrow=8)
                    imageio.save_image(grid, os.path.join(temp_results_dir, 'div_iter_{}.jpg'.format(iterations // 1000)))
                if iterations % 20000 == 1:
                    torch.save(generator.state_dict(), models_dir + '/' + args.model_name + 'G_' + str(iterations // 15000))
                    torch.save(discriminator.state_dict(), models_dir + '/' + args.model_name + 'D_' + str(iterations // 15000))


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
