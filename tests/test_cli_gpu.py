"""End to end through the CLIs on the GPU: `train_model.main([...])` on a 4-image folder (2 epochs x 2 iterations: a generator
step, critic steps, a captured-and-replayed iteration, checkpoints, temp-result grids, event file) and `eval_model.main([...])`
on the checkpoint it wrote -- the output files and directories of reference train_model.py:281-290,347-376,402-408 and
eval_model.py:58-87."""
import os

import numpy as np
import pytest
import torch

from oracle import recipes

pytestmark = pytest.mark.gpu


def test_train_and_eval_cli(dev, tmp_path, monkeypatch):
    from PIL import Image
    from semantic_pyramid_for_image_generation_amd import eval_model, imageio, train_model
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(1)
    for cls in ('park', 'street'):
        os.makedirs(tmp_path / 'data' / cls)
        for i in range(2):
            Image.fromarray(rng.randint(0, 255, (270, 300, 3), dtype=np.uint8)).save(tmp_path / 'data' / cls / f'{i}.jpg')
    torch.save(recipes.make_resnet(2).state_dict(), tmp_path / 'classifier18')
    torch.manual_seed(0)
    train_model.main(['--model-name', 'm', '--train-path', str(tmp_path / 'data'), '--classifier-path', str(tmp_path / 'classifier18'),
                      '--batch-size', '2', '--epochs', '3', '--keep-temp-results', '--fixed-layer', '2'])
    torch.cuda.synchronize()
    root = tmp_path / 'wgan-gp_models' / 'm'
    hp = open(root / 'hyperparams.txt').read()
    assert 'batch_size: 2' in hp and 'discriminator_norm: instance_norm' in hp and 'gen_type: res' in hp
    assert sorted(os.listdir(root / 'models_checkpoint')) == ['mD', 'mD_0', 'mG', 'mG_0']
    assert sorted(os.listdir(root / 'temp_results')) == ['div_iter_0.jpg', 'original_images.jpg', 'original_images_diversity.jpg',
                                                        'res_iter_0.jpg']
    assert Image.open(root / 'temp_results' / 'res_iter_0.jpg').size == (8 * 258 + 2, 258 + 2)         # 4 levels x batch 2 in one row of 8
    ev = os.listdir(root / 'summaries')
    assert len(ev) == 1
    scalars = imageio.read_event_scalars(str(root / 'summaries' / ev[0]))
    tags = {t for _, t, _ in scalars}
    assert {'D/d_loss', 'D/gp', 'G/g_loss', 'G/features_loss', 'G/total_loss'} <= tags
    assert all(np.isfinite(v) for _, _, v in scalars)
    sd = torch.load(root / 'models_checkpoint' / 'mG', map_location='cpu', weights_only=True)
    Generator('res').load_state_dict(sd, strict=True)
    assert int(sd['conv_block1.block.1.num_batches_tracked']) >= 1

    eval_model.main(['--generator-path', str(root / 'models_checkpoint' / 'mG'), '--eval-path', str(tmp_path / 'data'),
                     '--batch-size', '2', '--num-of-samples', '1', '--output-path', str(tmp_path / 'out'),
                     '--classifier-path', str(tmp_path / 'classifier18')])
    out = tmp_path / 'out' / 'mG'
    assert sorted(os.listdir(out)) == ['orig_1.jpg'] + ['sample_1_f%d.jpg' % l for l in range(1, 5)]
    assert Image.open(out / 'sample_1_f3.jpg').size == (8 * 258 + 2, 258 + 2)                          # 4 noises x batch 2 in one row
