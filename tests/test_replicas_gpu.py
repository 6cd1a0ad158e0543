"""The data-parallel oracle of SURVEY.md §8(e) on one GPU: N = 2 replicas with identical parameters and rank-distinct data
go through the REAL step functions (train_discriminator / train_generator) with flat gradients; the collective is stood in
for by a reducer that sums the two replicas' flat buffers (exactly what the all-reduce leaves in the buffer), the optimizer
divides by N (grad_scale) and leaves zeros behind.  Post-step parameters must equal two reference replicas (the CPU
oracle, fp64) whose gradients are averaged before Adam."""
import random
from argparse import Namespace

import pytest
import torch

from golden_util import cpu_state
from oracle import masks as omasks
from oracle import nets, recipes, steps

pytestmark = pytest.mark.gpu


class _Stop(Exception):
    pass


def _models(dev, seed, gen_type='res', norm='instance_norm'):
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    torch.manual_seed(seed)
    G = Generator(gen_type); G.init_weights()
    D = Discriminator(norm, gen_type); D.init_weights()
    csd = {k: v.detach().clone() for k, v in recipes.make_resnet(seed).state_dict().items()}
    C = Classifier(resnet18(365)); C.load_state_dict(csd)
    return G, D, C, csd


def test_two_replicas_match_averaged_reference(dev):
    from semantic_pyramid_for_image_generation_amd import optim, parallel
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    seed, B, L = 33, 2, 2
    G, D, C, csd = _models(dev, seed)
    gsd, dsd = cpu_state(G), cpu_state(D)
    tm.args = Namespace(use_diversity_loss=False, gradient_penalty_weight=10.0)
    d_loss_fn, g_loss_fn = tm.get_wgan_losses_fn()
    crit = tm.L1Loss()
    # rank-distinct data, identical mask draw (same Python-random seed on every rank)
    data = []
    for rank in range(2):
        images = recipes.image_batch(seed + 10 * rank, B)
        with torch.no_grad():
            _, feats = nets.classifier_forward(csd, recipes.normalize_clf(images))
        ms, _ = omasks.masks_part2([tuple(f.shape) for f in feats], L, random.Random(seed))
        gen = torch.Generator().manual_seed(seed + 100 + rank)
        data.append(dict(images_d=(images - 0.5) / 0.5, feats=feats, masks=[torch.from_numpy(m) for m in ms],
                         z=torch.randn(B, 128, 1, 1, generator=gen), alpha=torch.rand(B, 1, 1, 1, generator=gen),
                         zg=torch.randn(B, 128, 1, 1, generator=gen)))

    # ---- reference: two CPU replicas (fp64), gradients averaged, one Adam step each for D and G
    f64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    d64 = lambda t: [x.double() for x in t] if isinstance(t, list) else t.double()
    reps = [steps.Trainer(f64(gsd), f64(dsd), f64(csd), 'res', 'instance_norm') for _ in range(2)]
    for which in ('d', 'g'):
        grads = []
        for r, dt in zip(reps, data):
            if which == 'd':
                fake = r.G(d64(dt['z']), d64(dt['feats']), d64(dt['masks'])).detach()
                rl, fl = -r.D(d64(dt['images_d'])).mean(), r.D(fake).mean()
                loss = (rl + fl) + steps.gradient_penalty(r.D, d64(dt['images_d']), fake, d64(dt['alpha'])) * 10.0
                params = [r.d[k] for k in r.d_params]
            else:
                fake = r.G(d64(dt['zg']), d64(dt['feats']), d64(dt['masks']))
                _, ff = nets.classifier_forward(r.clf, steps.normalize(fake, steps.CLF_MEAN, steps.CLF_STD))
                loss = -r.D(fake).mean() + steps.features_loss(d64(dt['feats']), d64(dt['masks']), False, L, ff)
                params = [r.g[k] for k in r.g_params]
            grads.append(torch.autograd.grad(loss, params, allow_unused=True))
        for r in reps:                                # every replica applies the SAME averaged gradient
            params = [r.d[k] for k in r.d_params] if which == 'd' else [r.g[k] for k in r.g_params]
            for i, p in enumerate(params):
                p.grad = None if grads[0][i] is None else (grads[0][i] + grads[1][i]) / 2
            (r.d_opt if which == 'd' else r.g_opt).step()
    want_d = {k: reps[0].d[k].detach() for k in reps[0].d_params}
    want_g = {k: reps[0].g[k].detach() for k in reps[0].g_params}

    # ---- HIP path: replica 1 runs up to the collective and hands over its buffer; replica 0 receives the sum
    def replica(rank):
        G_, D_, C_, _ = _models(dev, seed)
        G_.to(dev); D_.to(dev); C_.to(dev).eval()
        for p in C_.parameters():
            p.requires_grad_(False)
        parallel.flatten_grads(G_); parallel.flatten_grads(D_)
        d_opt = optim.Adam(D_.parameters(), lr=2e-4, betas=(0.5, 0.999), grad_scale=0.5, zero_grads=True)
        g_opt = optim.Adam(G_.parameters(), lr=2e-4, betas=(0.5, 0.999), grad_scale=0.5, zero_grads=True)
        dt = data[rank]
        to = lambda t: [x.to(dev) for x in t] if isinstance(t, list) else t.to(dev)
        return G_, D_, C_, d_opt, g_opt, {k: to(v) for k, v in dt.items()}

    G0, D0, C0, d0, g0, t0 = replica(0)
    G1, D1, C1, d1, g1, t1 = replica(1)
    other = {}

    def run_d(G_, D_, opt, t):
        return tm.train_discriminator(G_, D_, d_loss_fn, opt, t['images_d'], t['feats'], t['masks'], z=t['z'], alpha=t['alpha'])

    def run_g(G_, D_, C_, opt, t):
        return tm.train_generator(G_, D_, g_loss_fn, opt, B, t['feats'], crit, L, C_, tm.Normalize(recipes.CLF_MEAN, recipes.CLF_STD),
                                  crit, crit, t['masks'], [2], z=t['zg'])

    def capture(flat):
        other['flat'] = flat.clone()
        raise _Stop()

    def reduce_sum(flat):
        flat.add_(other['flat'])

    try:
        for step in ('d', 'g'):
            parallel._REDUCER = capture
            with pytest.raises(_Stop):
                run_d(G1, D1, d1, t1) if step == 'd' else run_g(G1, D1, C1, g1, t1)
            parallel._REDUCER = reduce_sum
            run_d(G0, D0, d0, t0) if step == 'd' else run_g(G0, D0, C0, g0, t0)
            # replica 1 applies the same update (its generator step must see the same critic): copy replica 0's parameters
            with torch.no_grad():
                for a, b in zip(D1.parameters(), D0.parameters()):
                    a.copy_(b)
                for a, b in zip(G1.parameters(), G0.parameters()):
                    a.copy_(b)
            for m in (D1, G1):                       # replica 1 stopped before its optimizer step: clear what it accumulated
                m._sempyr_flat_grads.flat.zero_()
            from semantic_pyramid_for_image_generation_amd import ops
            for opt_ in (d1, g1):
                for grp in opt_.param_groups:
                    torch.autograd.graph.increment_version(list(grp['params']))
                    ops.repack_cached(grp['params'], None)
    finally:
        parallel._REDUCER = None
    torch.cuda.synchronize()
    # the flat buffers are zero again (zero_grads) and still are the gradients
    assert float(D0._sempyr_flat_grads.flat.abs().max()) == 0.0 and float(G0._sempyr_flat_grads.flat.abs().max()) == 0.0
    assert all(p.grad is v for p, v in zip(D0._sempyr_flat_grads.params, D0._sempyr_flat_grads.views))
    # one Adam step moves every element by ~lr: compare the UPDATE with the reference's (sign flips of round-off gradients allowed)
    lr = 2e-4
    for name, mod, want, init in (('D', D0, want_d, dsd), ('G', G0, want_g, gsd)):
        checked = 0
        for k, p in mod.named_parameters():
            ref_delta = want[k].double() - init[k].double()
            if float(ref_delta.abs().max()) < 0.5 * lr:
                continue
            delta = p.detach().double().cpu() - init[k].double()
            diff = (delta - ref_delta).abs() / lr
            assert float(diff.median()) <= 0.05, (name, k, float(diff.median()))
            assert float((diff > 0.5).double().mean()) <= 0.1, (name, k, float((diff > 0.5).double().mean()))
            checked += 1
        assert checked >= 10
