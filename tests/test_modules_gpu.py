"""Module-level parity on the GPU: the HIP-backed drop-in modules against the CPU oracle (same state_dict,
same seeded inputs) and against the reference-generated golden fixtures.  Tolerance: max-abs 1e-3 on images
and logits (BASELINE.json north_star, fp32); observed errors are ~1e-5."""
import os
import random

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, check_summary, cpu_state, load_json, load_npz, param_checksum
from oracle import masks as omasks
from oracle import nets, recipes

pytestmark = pytest.mark.gpu
TOL = 1e-3


def maxabs(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def to_dev(ts, dev):
    return [t.to(dev) for t in ts]


@pytest.mark.parametrize("gen_type", ['res', 'default'])
@pytest.mark.parametrize("mode", ['train', 'eval'])
def test_generator(dev, gen_type, mode):
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    store, meta = load_npz('modules.npz'), load_json('modules.json')
    name = f'G_{gen_type}_{mode}'
    seed = meta[name]['seed']
    torch.manual_seed(seed)
    g = Generator(gen_type)
    g.init_weights()
    assert abs(param_checksum(g) - meta[name]['param_checksum']) < 1e-6 * meta[name]['param_checksum']
    z, feats, masks = recipes.generator_inputs(seed, 2)
    if mode == 'eval':
        recipes.perturb_bn_buffers(g, seed)
    sd = cpu_state(g)
    with torch.no_grad():
        want = nets.generator_forward(sd, gen_type, z, feats, masks, training=(mode == 'train'))
    g.to(dev)
    g.train(mode == 'train')
    with torch.no_grad():
        got = g(z.to(dev), to_dev(feats, dev), to_dev(masks, dev))
    assert maxabs(got, want) < TOL
    check_summary(store, name, got, TOL)
    if mode == 'train':   # BatchNorm running buffers were updated like the reference's
        bn = g.conv_block2.block[1]
        assert np.allclose(bn.running_mean.cpu().numpy(), store[name + '.rm'], atol=1e-5)
        assert np.allclose(bn.running_var.cpu().numpy(), store[name + '.rv'], atol=1e-5)
        assert int(bn.num_batches_tracked) == meta[name]['nbt']
        for k, v in g.state_dict().items():
            assert maxabs(v, sd[k]) < 1e-4, k


def test_generator_detached_forward_equals_autograd_forward(dev):
    """The no-grad generator forward (fused norm-apply-into-conv staging, statistics from conv epilogues) gives bit for bit
    the image and the BatchNorm running statistics of the autograd forward (the unfused formulation)."""
    import copy
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    torch.manual_seed(5)
    g1 = Generator('res'); g1.init_weights(); g1.to(dev).train()
    g2 = copy.deepcopy(g1)
    z, feats, masks = recipes.generator_inputs(5, 2)
    z, feats, masks = z.to(dev), to_dev(feats, dev), to_dev(masks, dev)
    with torch.no_grad():
        a = g1(z, feats, masks)
    b = g2(z, feats, masks)
    assert b.requires_grad and torch.equal(a, b.detach())
    for (k, u), (_, v) in zip(g1.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(u, v), k


@pytest.mark.parametrize("norm", ['instance_norm', 'batch_norm', 'layer_norm'])
@pytest.mark.parametrize("dis_type", ['res', 'default'])
def test_discriminator(dev, norm, dis_type):
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    store, meta = load_npz('modules.npz'), load_json('modules.json')
    name = f'D_{norm}_{dis_type}'
    seed = meta[name]['seed']
    torch.manual_seed(seed)
    d = Discriminator(norm, dis_type)
    d.init_weights()
    assert abs(param_checksum(d) - meta[name]['param_checksum']) < 1e-6 * meta[name]['param_checksum']
    x = recipes.image_batch(seed, 2) * 2 - 1
    d.to(dev).train()
    with torch.no_grad():
        got = d(x.to(dev))
    assert got.shape == (2, 1, 1, 1)
    assert np.allclose(got.cpu().numpy().reshape(-1), store[name], atol=TOL)


def test_feature_models(dev):
    from semantic_pyramid_for_image_generation_amd.featuresCreatorGen import (Features1ToImage, FeaturesDiscriminator,
                                                                               LevelUpFeaturesGenerator)
    store, meta = load_npz('modules.npz'), load_json('modules.json')
    f1 = Features1ToImage()
    f1.load_state_dict(torch.load(os.path.join(GOLDEN, 'features1_to_image.pt'), weights_only=True), strict=True)
    f1.to(dev).eval()
    g = torch.Generator().manual_seed(meta['F1']['seed'])
    x = torch.rand(2, 64, 64, 64, generator=g) * 2.0
    with torch.no_grad():
        y = f1(x.to(dev))
    check_summary(store, 'F1', y, TOL)          # real trained weights shipped with the reference
    shapes = recipes.tap_shapes(2)
    for level in [2, 3, 4]:
        seed = meta[f'LU{level}']['seed']
        torch.manual_seed(seed)
        lu = LevelUpFeaturesGenerator(level)
        lu.init_weights()
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(shapes[level], generator=g)
        lu.to(dev).train()
        with torch.no_grad():
            y = lu(x.to(dev))
        check_summary(store, f'LU{level}', y, TOL)
        for norm in ['instance_norm', 'batch_norm', 'layer_norm']:
            for dis_type in (['res', 'default'] if norm == 'instance_norm' else ['res']):
                torch.manual_seed(seed + 100)
                fd = FeaturesDiscriminator(norm, dis_type, level - 1)
                fd.init_weights()
                xx = torch.rand(shapes[level - 1], generator=g)
                fd.to(dev).train()
                with torch.no_grad():
                    yy = fd(xx.to(dev))
                assert np.allclose(yy.cpu().numpy().reshape(-1), store[f'FD{level - 1}_{norm}_{dis_type}'], atol=TOL), \
                    (level, norm, dis_type)


def test_classifier_taps(dev):
    """ResNet-18 taps vs the oracle restatement ("parity unpinned" by reference data: torchvision and the trained
    weights are absent; seeded random weights of the same architecture)."""
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    model = recipes.make_resnet(7)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    clf = Classifier(resnet18(365))
    clf.load_state_dict(sd, strict=True)
    clf.to(dev).eval()
    x = recipes.normalize_clf(recipes.image_batch(7, 2))
    with torch.no_grad():
        want_logits, want_taps = nets.classifier_forward(sd, x)
        logits, taps = clf(x.to(dev))
    assert maxabs(logits, want_logits) < TOL
    for i, (a, b) in enumerate(zip(taps, want_taps)):
        assert a.shape == b.shape
        assert maxabs(a, b) < TOL, i
    # layer4 called directly, as train_feature_creation_gen.py:46 does
    with torch.no_grad():
        l4 = clf.layer4(want_taps[3].to(dev))
    assert maxabs(l4, want_taps[4]) < TOL


def test_eval_loop(dev):
    """eval_model.py inner loop (one level, injected noises) against the oracle."""
    from semantic_pyramid_for_image_generation_amd import eval_model
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    from semantic_pyramid_for_image_generation_amd.train_model import Normalize
    torch.manual_seed(3)
    g = Generator('res'); g.init_weights(); recipes.perturb_bn_buffers(g, 3)
    gsd = cpu_state(g)
    model = recipes.make_resnet(3)
    csd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    clf = Classifier(resnet18(365)); clf.load_state_dict(csd); clf.to(dev).eval()
    g.to(dev).eval()
    images = recipes.image_batch(3, 2)
    gen = torch.Generator().manual_seed(99)
    noises = [[torch.randn(2, 128, 1, 1, generator=gen) for _ in range(2)] for _ in range(4)]
    outs = eval_model.synthesize_batch(g, clf, images.to(dev), Normalize(recipes.CLF_MEAN, recipes.CLF_STD),
                                       noises=[[z.to(dev) for z in lv] for lv in noises], num_noises=2)
    with torch.no_grad():
        _, feats = nets.classifier_forward(csd, recipes.normalize_clf(images))
        masks = [torch.ones_like(f) for f in feats]
        for level in (1, 4):
            fl = [f if i == level else f * 0 for i, f in enumerate(feats)]
            want = torch.vstack([nets.generator_forward(gsd, 'res', z, fl, masks, False) for z in noises[level - 1]])
            assert maxabs(outs[level], want) < TOL


def test_masks_on_device_bit_exact(dev):
    from semantic_pyramid_for_image_generation_amd import train_model
    shapes = recipes.tap_shapes(2)
    for seed in range(3):
        for L in range(1, 5):
            random.seed(seed)
            masks = [torch.full(s, 0.5, device=dev) for s in shapes]
            train_model.setMasksPart2(masks, dev, L)
            want, _ = omasks.masks_part2(shapes, L, random.Random(seed))
            for m, w in zip(masks, want):
                assert np.array_equal(m.cpu().numpy(), w)
