"""The captured loop body (loop.TrainLoop) against eager execution of the same body, and the order of its random draws.

  * replay (sempyr_replay_launch: plain stream launches re-issued from native code) and hipGraphLaunch of the captured body
    are BIT-identical to issuing the body from Python, over 7 iterations (two generator steps, every layer variant that the
    seeded draws select), including `prepare()`'s throw-away warm-up runs being undone exactly;
  * without injected noise the loop draws z (CPU generator), alpha and the generator step's z (device generator) in the
    reference's order (train_model.py:128, :53, :90), and train type / layer / crop from Python's `random` in the
    reference's order (:301, :327, :201)."""
import random
from argparse import Namespace

import pytest
import torch

from oracle import masks as omasks
from oracle import recipes

pytestmark = pytest.mark.gpu


def _make(dev, use_graphs, executor='own', B=4, norm='instance_norm', gen_type='res', div=False, seed=11, depth=None):
    from semantic_pyramid_for_image_generation_amd import loop, optim
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    torch.manual_seed(seed)
    G = Generator(gen_type); G.init_weights()
    D = Discriminator(norm, gen_type); D.init_weights()
    C = Classifier(resnet18(365)); C.load_state_dict(recipes.make_resnet(seed).state_dict())
    G.to(dev); D.to(dev); C.to(dev).eval()
    for p in C.parameters():
        p.requires_grad_(False)
    tm.args = Namespace(use_diversity_loss=div, gradient_penalty_weight=10.0)
    g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    return G, D, loop.TrainLoop(G, D, C, g_opt, d_opt, B, dev, use_graphs=use_graphs, executor=executor, lookahead=depth)


def _run(dev, use_graphs, executor, iters, prepare, lookahead=False, **kw):
    G, D, lp = _make(dev, use_graphs, executor, **kw)
    pool = [recipes.image_batch(40 + i, lp.B).to(dev) for i in range(3)]
    if prepare:
        lp.prepare()
    torch.manual_seed(7); random.seed(7)
    for i in range(iters):
        nxt = [pool[j % 3] for j in range(i + 1, min(i + 1 + lp.lookahead, iters))] if lookahead else None
        lp.iteration(pool[i % 3], log=True, next_images=nxt or None)
    logged = lp.pop_logged()
    torch.cuda.synchronize()
    state = {('G', k): v.detach().clone() for k, v in G.state_dict().items()}
    state.update({('D', k): v.detach().clone() for k, v in D.state_dict().items()})
    return logged, state, lp


@pytest.mark.parametrize("executor", ["own", "hip"])
def test_replay_bit_identical_to_eager(dev, executor):
    le, se, _ = _run(dev, False, 'own', 7, False)
    lg, sg, lp = _run(dev, True, executor, 7, True)
    assert len(lp.graphs) == 8 * lp.nslots                      # (generator step or not) x layer 1..4 x buffer set, all captured by prepare()
    if executor == 'own':
        stats = lp.graphs[((False, False, 1), 0)].stats
        assert stats[0] > 200 and stats[2] == 0 and stats[4] >= 2      # kernels; no memcpy nodes; more than one stream
    assert [it for it, _ in le] == list(range(1, 8))
    for (ia, a), (ib, b) in zip(le, lg):
        assert ia == ib and a == b, (ia, a, b)
    assert sum(1 for _, d in le if 'G/g_loss' in d) == 2        # iterations 1 and 6 end with a generator step
    for k in se:
        assert torch.equal(se[k], sg[k]), k


def test_lookahead_launch_order_changes_nothing(dev):
    """iteration(next_images=[...]): stage A of the next 1..3 iterations is launched ahead of stage B of iteration i (never
    across a generator step); 12 iterations with three generator steps, prepared and captured-on-the-fly: same bits as
    all-eager"""
    le, se, _ = _run(dev, False, 'own', 12, False)
    for prepare, depth in ((True, 1), (True, 2), (False, 2), (True, 3)):
        lg, sg, lp = _run(dev, True, 'own', 12, prepare, lookahead=True, depth=depth)
        assert lp._ahead == [] and lp.nslots == depth + 1
        assert [d for _, d in le] == [d for _, d in lg]
        assert [it for it, _ in lg] == list(range(1, 13))
        for k in se:
            assert torch.equal(se[k], sg[k]), (prepare, k)


def test_replay_without_prepare_captures_on_second_occurrence(dev):
    """a variant runs eagerly when first met and is captured when met again: same results as all-eager"""
    le, se, _ = _run(dev, False, 'own', 14, False, B=2, norm='layer_norm')
    lg, sg, lp = _run(dev, True, 'own', 14, False, B=2, norm='layer_norm')
    assert 1 <= len(lp.graphs) <= 7          # (variant, buffer set) pairs met at least twice in 14 iterations
    assert [d for _, d in le] == [d for _, d in lg]
    for k in se:
        assert torch.equal(se[k], sg[k]), k


def test_replay_batch_norm_critic_and_diversity(dev):
    """BatchNorm critic (running statistics and num_batches_tracked are written by the replayed kernels) + diversity loss"""
    le, se, _ = _run(dev, False, 'own', 6, False, B=2, norm='batch_norm', gen_type='default', div=True)
    lg, sg, _ = _run(dev, True, 'own', 6, True, B=2, norm='batch_norm', gen_type='default', div=True)
    assert [d for _, d in le] == [d for _, d in lg]
    assert any('G/diversity loss' in d for _, d in le)
    for k in se:
        assert torch.equal(se[k], sg[k]), k
    assert int(se[('D', 'conv_block1.conv_block1.block.1.num_batches_tracked')]) == 6 * 3 + 2     # 3 critic passes per D step + 1 per G step (iterations 1 and 6)


def test_exact_shortcuts_change_nothing(dev):
    """The two shortcuts that rest on the reference's all-zero masks (quirk Q2) -- no convolution of a zero-masked tap
    (generator._ZeroConv) and no classifier layer behind the deepest tap with a non-zero mask (train_model.live_taps) -- are
    exact: 11 iterations (three generator steps, every layer 1..4 drawn) with and without them give the same logged losses
    and bit-identical generator / critic parameters and BatchNorm buffers."""
    from semantic_pyramid_for_image_generation_amd import generator as gmod
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    saved = (gmod.ZERO_SKIP, tm.DEAD_TAPS)
    try:
        gmod.ZERO_SKIP, tm.DEAD_TAPS = True, True
        l1, s1, lp = _run(dev, False, 'own', 11, False, B=2)
        gmod.ZERO_SKIP, tm.DEAD_TAPS = False, False
        l0, s0, _ = _run(dev, False, 'own', 11, False, B=2)
    finally:
        gmod.ZERO_SKIP, tm.DEAD_TAPS = saved
    assert [d for _, d in l1] == [d for _, d in l0]
    assert sum(1 for _, d in l1 if 'G/g_loss' in d) == 3
    for k in s1:
        assert torch.equal(s1[k], s0[k]), k


def test_live_taps_and_partial_classifier(dev):
    """Classifier.forward(x, taps_upto=k) returns exactly the taps of the full forward up to layer k and None behind it;
    live_taps reads the deepest non-zero mask from the tags of the mask setters."""
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    G, D, lp = _make(dev, False, B=2)
    x = recipes.image_batch(5, 2).to(dev)
    with torch.no_grad():
        _, full = lp.C(lp.normalizer_clf(x))
        for k in range(0, 5):
            out, taps = lp.C(lp.normalizer_clf(x), taps_upto=k)
            assert out is None and taps[0] is None and taps[5] is None
            for i in range(1, 5):
                assert (taps[i] is None) if i > k else torch.equal(taps[i], full[i])
    random.seed(3)
    seen = set()
    for _ in range(40):
        key = lp.draw()
        masks = [None] + [tm.tag_mask(torch.empty(0, device=dev), lp.specs[i]) for i in range(1, 5)] + [None]
        assert tm.live_taps(masks) == key[2] - 1                 # quirk Q2: the masks at and above the drawn layer are zero
        seen.add(key[2])
    assert seen == {1, 2, 3, 4}


def test_draw_order_matches_reference(dev):
    G, D, lp = _make(dev, False, B=3)
    images = recipes.image_batch(3, 3).to(dev)
    torch.manual_seed(123); random.seed(123)
    key1 = lp.draw()                                            # iteration 1: critic step + generator step
    i1 = lp._draw_device(key1, images, None)
    got1 = (i1.z_d.clone(), i1.alpha.clone(), i1.z_g.clone(), lp.train_type, lp.features_to_train, list(lp.specs))
    key2 = lp.draw()                                            # iteration 2: critic step only
    i2 = lp._draw_device(key2, images, lp._slots()[0])
    assert i2.z_g is lp.slots[0].z_g
    got2 = (i2.z_d.clone(), i2.alpha.clone(), lp.train_type, lp.features_to_train, list(lp.specs))
    # the reference's order, written out
    torch.manual_seed(123); rnd = random.Random(123)
    shapes = recipes.tap_shapes(3)
    is1, L1 = omasks.draw_iteration(rnd, 0.6)
    m1, crop1 = omasks.masks_part2(shapes, L1, rnd)
    z_d1 = torch.randn(3, 128, 1, 1)                            # train_model.py:128 (CPU generator)
    alpha1 = torch.rand(3, 1, 1, 1, device=dev)                 # :53
    z_g1 = torch.randn(3, 128, 1, 1, device=dev)                # :90
    is2, L2 = omasks.draw_iteration(rnd, 0.6)
    m2, crop2 = omasks.masks_part2(shapes, L2, rnd)
    z_d2 = torch.randn(3, 128, 1, 1)
    alpha2 = torch.rand(3, 1, 1, 1, device=dev)
    assert key1 == (True, False, L1) and key2 == (False, False, L2)
    assert torch.equal(got1[0].cpu(), z_d1) and torch.equal(got1[1], alpha1) and torch.equal(got1[2], z_g1)
    assert torch.equal(got2[0].cpu(), z_d2) and torch.equal(got2[1], alpha2)
    assert (got1[3] == 1) is False and got1[4] == L1 and got2[3] == L2
    # the specs describe exactly the oracle's masks
    from semantic_pyramid_for_image_generation_amd import ops
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    for specs, want in ((got1[5], m1), (got2[4], m2)):
        masks = [torch.empty(s, device=dev) for s in shapes]
        spec_dev = torch.empty(36, dtype=torch.int32, device=dev)
        ops.store_words(spec_dev, tm.spec_words(specs))
        ops.build_masks(masks, spec_dev)
        for m, w in zip(masks, want):
            assert torch.equal(m.cpu(), torch.from_numpy(w))


def test_norm_is_twice_differentiable(dev):
    """train_model.norm (reference train_model.py:58-60) composed by hand into the penalty, as the reference's one_mean_gp
    does under create_graph=True: forward, backward and double backward against torch."""
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(4, 3, 8, 8, generator=g).to(dev)
    w0 = torch.randn(4, 3, 8, 8, generator=g).to(dev)

    def penalty(norm_fn):
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        y = (x * w).tanh()                                       # something with a graph, like d pred / d x
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        n = norm_fn(gx)
        gp = ((n - 1) ** 2).mean()
        gp.backward()
        return n.detach(), gp.detach(), w.grad.detach()
    n1, gp1, gw1 = penalty(tm.norm)
    n0, gp0, gw0 = penalty(lambda t: t.view(t.size(0), -1).norm(p=2, dim=1))
    assert torch.allclose(n1, n0, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gp1, gp0, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gw1, gw0, rtol=1e-4, atol=1e-6)
    z = torch.zeros(2, 5, device=dev, requires_grad=True)       # zero rows: zero gradient, not NaN
    tm.norm(z).sum().backward()
    assert torch.equal(z.grad, torch.zeros_like(z))
