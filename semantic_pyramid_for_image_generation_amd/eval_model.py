"""Synthesis CLI — drop-in mirror of the reference `eval_model.py`: for every source batch, classifier
taps -> for each level 1..4 keep only that level's features -> 4 noises -> generator (eval mode) -> grids.
"""
import argparse
import os

import torch

from . import ops, parallel
from .generator import Generator
from .train_model import Normalize, load_classifier

args = None


def one_level(features, level):
    """clone the taps and zero every level but `level` (reference eval_model.py:72-75)."""
    out = []
    for i, f in enumerate(features):
        out.append(torch.clone(f) if i == level else torch.zeros_like(f))
    return out


def synthesize_batch(generator, classifier, images, normalizer_clf, num_levels=4, noises=None, num_noises=4):
    """The hot loop of reference eval_model.py:64-85 for one source batch: returns
    {level: (num_noises*B, 3, 256, 256)}.  `noises[level-1][i]` may be injected; otherwise z ~ N(0,1) on the device."""
    device = images.device
    B = images.shape[0]
    with torch.no_grad():
        generator.eval()
        _, features = classifier(normalizer_clf(images))
        # eval-mode BatchNorm: samples are independent, so the num_noises generator calls of the reference loop run as
        # ONE batch of num_noises*B (same images in the same order, 4x fewer and 4x fuller launches).  The batched masks
        # (all ones), the batched taps and the batched zero taps of the other levels (eval_model.py:72-75) are the same for
        # every level: they are built once per source batch, not once per level.
        rep = lambda t: t.repeat(num_noises, 1, 1, 1)
        masks = [torch.ones((num_noises * f.shape[0],) + tuple(f.shape[1:]), device=device) for f in features]
        taps = [rep(f) for f in features]
        zeros = [torch.zeros_like(t) for t in taps]
        out = {}
        for level in range(1, num_levels + 1):
            feats = [taps[i] if i == level else zeros[i] for i in range(len(features))]
            zs = [noises[level - 1][i] if noises is not None else torch.randn(B, 128, 1, 1, device=device)
                  for i in range(num_noises)]
            out[level] = generator(torch.cat(zs), feats, masks)
    return out


def _main():
    from . import imageio
    device = parallel.init_device()
    generator = Generator(gen_type=args.gen_type)
    generator.load_state_dict(torch.load(args.generator_path, map_location='cpu', weights_only=True))
    generator.eval()
    generator.to(device)
    classifier = load_classifier(args.classifier_path, device)
    classifier.eval()

    eval_set = imageio.ImageFolder(args.eval_path, image_size=256, crop_size=256, raw=imageio.DEVICE_TRANSFORM)
    normalizer_clf = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    print('set data loader')
    eval_loader = imageio.make_loader(eval_set, args.batch_size, shuffle=True, num_workers=4, drop_last=True)
    output_images_dir = os.path.join(args.output_path, args.generator_path.split('/')[-1])
    os.makedirs(output_images_dir, exist_ok=True)
    print("Starting Eval Loop...")
    sample_num = 1
    for images, _ in eval_loader:
        images = imageio.batch_images(images, device)
        grid = imageio.make_grid(images, padding=2, normalize=False, nrow=8)
        imageio.save_image(grid, os.path.join(output_images_dir, 'orig_{}.jpg'.format(sample_num)))
        # the reference parses --num-of-noises but always loops range(4) (quirk Q6)
        outs = synthesize_batch(generator, classifier, images, normalizer_clf, num_noises=4)
        for level, output_images in outs.items():
            grid = imageio.make_grid(output_images, padding=2, normalize=True, nrow=8)
            imageio.save_image(grid, os.path.join(output_images_dir, 'sample_{}_f{}.jpg'.format(sample_num, level)))
        sample_num += 1
        if sample_num == args.num_of_samples + 1:
            print("done")
            break


def build_parser():
    parser = argparse.ArgumentParser('choose hyperparameters')
    parser.add_argument('--generator-path', required=True)
    parser.add_argument('--eval-path', required=True)
    parser.add_argument('--batch-size', default=8, type=int)
    parser.add_argument('--num-of-samples', default=5, type=int)
    parser.add_argument('--num-of-noises', default=4, type=int)
    parser.add_argument('--output-path', required=True)
    parser.add_argument('--gen-type', default='res', choices=['default', 'res'])
    parser.add_argument('--classifier-path', default='/home/dcor/ronmokady/workshop21/team1/try/classifier18')
    return parser


def main(argv=None):
    global args
    args = build_parser().parse_args(argv)
    _main()


if __name__ == '__main__':
    main()
