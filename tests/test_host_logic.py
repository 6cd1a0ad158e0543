"""Host-side logic that needs no GPU: mask setters (bit-exact vs reference goldens), state_dict key grammar,
CLI flags/defaults, classifier checkpoint shim, image-grid helper, and the rule that the product never
touches the oracle or a CPU fallback."""
import hashlib
import os
import random
import re

import numpy as np
import pytest
import torch

from golden_util import load_json
from oracle import recipes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'semantic_pyramid_for_image_generation_amd')


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def test_product_masks_bit_exact_vs_reference():
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    gold = load_json('masks.json')
    shapes = recipes.tap_shapes(2)
    cpu = torch.device('cpu')
    for case in gold['cases']:
        random.seed(case['seed'])
        tt = random.choices([1, 2], [0.6, 0.4])
        assert (tt == 1) is False
        masks = [torch.full(s, 0.5) for s in shapes]
        tm.setMasksPart2(masks, cpu, case['L'])
        assert random.random() == case['next_random']
        assert [sha(m) for m in masks] == case['sha256']
    for c in gold['part1']:
        masks = [torch.full(s, 0.5) for s in shapes]
        tm.setMasksPart1(masks, cpu, c['L'])
        assert [sha(m) for m in masks] == c['sha256']
    g = torch.Generator().manual_seed(gold['isolate']['seed'])
    feats = [torch.rand(s, generator=g) for s in recipes.tap_shapes(1)]
    assert [sha(m) for m in tm.isolate_layer(feats, gold['isolate']['selected'], cpu)] == gold['isolate']['sha256']
    c = gold['cropped']
    m = tm.setCroppedMask(tuple(c['crop']), torch.ones(c['shape']), c['w1'], c['w2'], c['data'])
    assert sha(m) == c['sha256']


def test_state_dict_grammar():
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    g = Generator('res').state_dict()
    assert len(g) == 128 and sum(v.numel() for k, v in g.items() if v.is_floating_point() and 'running' not in k) == 9486403
    assert 'conv_block3.conv_block1.skip_block.block.1.num_batches_tracked' in g and g['last_conv.weight'].shape == (3, 64, 3, 3)
    gd = Generator('default').state_dict()
    assert len(gd) == 62 and gd['conv_block1.block.0.weight'].shape == (128, 512, 4, 4)      # ConvTranspose: (Cin, Cout, kh, kw)
    d = Discriminator('instance_norm', 'res').state_dict()
    assert len(d) == 24 and all(re.match(r'(first_conv|last_conv|conv_block[1-5]\.conv_block1\.block\.[01])\.(weight|bias)$', k) for k in d)
    assert len(Discriminator('batch_norm', 'default').state_dict()) == 39
    with pytest.raises(NotImplementedError):
        Discriminator('spectral', 'res')


def test_cli_flags_match_reference_defaults():
    from semantic_pyramid_for_image_generation_amd import eval_model, train_feature_creation_gen, train_model
    a = train_model.build_parser().parse_args(['--model-name', 'm'])
    assert (a.batch_size, a.lr, a.epochs, a.discriminator_norm, a.gradient_penalty_weight, a.discriminator_steps, a.gen_type,
            a.train1_prob, a.fixed_layer, a.keep_temp_results, a.use_diversity_loss) == \
        (16, 0.0002, 100, 'instance_norm', 10.0, 5, 'res', 0.6, None, False, False)
    e = eval_model.build_parser().parse_args(['--generator-path', 'g', '--eval-path', 'e', '--output-path', 'o'])
    assert (e.batch_size, e.num_of_samples, e.num_of_noises, e.gen_type) == (8, 5, 4, 'res')
    f = train_feature_creation_gen.build_parser().parse_args(['--features-gens-dir-path', 'd'])
    assert (f.batch_size, f.epochs, f.train_block_input) == (64, 500, 3)


def test_no_cpu_fallback_and_no_oracle_in_product():
    from semantic_pyramid_for_image_generation_amd import _lib, ops
    with pytest.raises(_lib.SempyrError):
        ops.conv2d(torch.randn(1, 3, 8, 8), torch.randn(4, 3, 3, 3), None, 1, 1)
    with pytest.raises(_lib.SempyrError):
        ops.norm_act(torch.randn(1, 3, 8, 8), torch.ones(3), torch.zeros(3), ops.NORM_INSTANCE)
    for dp, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{fn} imports the oracle'
                assert 'torch.nn.functional' not in src and 'F.conv2d' not in src, f'{fn} uses a torch compute fallback'


def test_make_grid_layout():
    from semantic_pyramid_for_image_generation_amd import imageio
    x = torch.arange(10 * 3 * 4 * 5, dtype=torch.float32).reshape(10, 3, 4, 5)
    grid = imageio.make_grid(x, nrow=8, padding=2, normalize=True)
    assert grid.shape == (3, 2 * (4 + 2) + 2, 8 * (5 + 2) + 2)
    assert float(grid.max()) == 1.0 and float(grid[:, :2].abs().max()) == 0.0
    assert torch.allclose(grid[:, 2:6, 2:7], (x[0] - x.min()) / (x.max() - x.min()))
    assert torch.allclose(grid[:, 8:12, 9:14], (x[9] - x.min()) / (x.max() - x.min()))


def test_loader_read_ahead_pairs():
    """train_model._with_next: every batch is prepared exactly once and handed to the loop together with the batches of the
    next calls (what TrainLoop.iteration(next_images=...) launches stage A for), in loader order"""
    from semantic_pyramid_for_image_generation_amd.train_model import _with_next
    seen = []

    def prep(x):
        seen.append(x)
        return x * 10

    assert list(_with_next(range(5), prep, 2)) == [(0, [10, 20]), (10, [20, 30]), (20, [30, 40]), (30, [40]), (40, None)]
    assert seen == [0, 1, 2, 3, 4]
    assert list(_with_next(range(3), prep, 1)) == [(0, [10]), (10, [20]), (20, None)]
    assert list(_with_next([7], prep, 3)) == [(70, None)]
    assert list(_with_next([], prep, 1)) == []
