"""Training of the per-level "features L -> features L-1" generators — drop-in mirror of the reference
`train_feature_creation_gen.py` (same step functions, schedule, flags, output files)."""
import argparse
import os
import time

import torch
from . import optim

from . import ops, parallel
from .featuresCreatorGen import Features1ToImage, FeaturesDiscriminator, LevelUpFeaturesGenerator
from .train_model import (L1Loss, Normalize, _critic_losses_batched, _ensure_train, _frozen, _one, _samplewise,
                          get_wgan_losses_fn, gradient_penalty, load_classifier, print_gpu_details)

args = None


def train_generator(features_gen, discriminator, classifier, features_gen_optimizer, features, features_to_train,
                    generator_loss_fn, criterion_features, next_level_features_criterion):
    """adversarial + classifier-stage reconstruction + 0.2 * next-level L1 (reference :30-58)."""
    _ensure_train(features_gen)
    _ensure_train(discriminator)
    with _frozen(discriminator, classifier):
        created = features_gen(features[features_to_train])
        generator_loss = generator_loss_fn(discriminator(created))
        if features_to_train == 2:
            reconstruct_features = classifier.layer2(created)
        elif features_to_train == 3:
            reconstruct_features = classifier.layer3(created)
        else:
            reconstruct_features = classifier.layer4(created)
        features_loss = criterion_features(features[features_to_train], reconstruct_features)
        next_level_features_loss = next_level_features_criterion(created, features[features_to_train - 1])
        losses_dictionary = {'generator_loss': generator_loss, 'features_loss': features_loss,
                             'next_level_loss': next_level_features_loss}
        total_loss = ops.lincomb([generator_loss, features_loss, next_level_features_loss], [1.0, 1.0, 0.2])
        parallel.zero_gradients(features_gen)
        with ops.async_param_grads():
            total_loss.backward(_one(created.device))
    parallel.sync_gradients(features_gen)
    features_gen_optimizer.step()
    return losses_dictionary


def train_discriminator(features_gen, discriminator, discriminator_loss_fn, discriminator_optimizer, features, features_to_train,
                        alpha=None):
    """critic step on feature tensors with the gradient penalty (reference :61-79)."""
    _ensure_train(features_gen)
    _ensure_train(discriminator)
    with torch.no_grad():
        fake_features = features_gen(features[features_to_train])
    real = features[features_to_train - 1]
    if _samplewise(discriminator):
        # no layer mixes samples: one critic pass over both batches (see train_model.train_discriminator)
        logits = discriminator(ops.concat_rows([real, fake_features]))
        real_loss, fake_loss = _critic_losses_batched(discriminator_loss_fn, logits, real.shape[0], fake_features.shape[0])
    else:
        real_loss, fake_loss = discriminator_loss_fn(discriminator(real), discriminator(fake_features))
    gp = gradient_penalty(discriminator, real, fake_features, alpha)
    discriminator_loss = ops.lincomb([real_loss, fake_loss, gp], [1.0, 1.0, args.gradient_penalty_weight])
    d_loss = ops.lincomb([real_loss.detach(), fake_loss.detach()])
    parallel.zero_gradients(discriminator)
    with ops.async_param_grads():
        discriminator_loss.backward(_one(real.device))
    parallel.sync_gradients(discriminator)
    discriminator_optimizer.step()
    return {'d_loss': d_loss, 'gp': gp}


def sample(f1_to_img, features_generator, features, features_level):
    """features_level -> ... -> level 1 -> image, everything in eval mode (reference :82-90)."""
    with torch.no_grad():
        for features_gen in features_generator:
            features_gen.eval()
        f1_to_img.eval()
        features_to_use = features[features_level]
        for i in range(features_level - 2, -1, -1):
            features_to_use = features_generator[i](features_to_use)
        return f1_to_img(features_to_use)


def _main():
    from . import imageio
    print_gpu_details()
    device = parallel.init_device()
    rank, world = parallel.rank(), parallel.world_size()
    print("set image folder")
    train_set = imageio.ImageFolder(args.train_path, image_size=256, crop_size=256, raw=imageio.DEVICE_TRANSFORM)
    normalizer_clf = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    print('set data loader')
    train_loader = imageio.make_loader(train_set, args.batch_size, shuffle=True, num_workers=4, drop_last=True,
                                       rank=rank, world_size=world)
    classifier = load_classifier(args.classifier_path, device)
    classifier.eval()
    for p in classifier.parameters():
        p.requires_grad_(False)
    features1_to_image_gen = Features1ToImage()
    features1_to_image_gen.load_state_dict(torch.load(os.path.join(args.features_gens_dir_path, 'features1_to_image'),
                                                      map_location='cpu', weights_only=True))
    features1_to_image_gen.eval()
    features1_to_image_gen.to(device)
    features_generators = [LevelUpFeaturesGenerator(input_level_features=i) for i in range(2, 5)]
    for i, features_gen in enumerate(features_generators):
        input_level_features = i + 2
        if input_level_features < args.train_block_input:
            features_gen.load_state_dict(torch.load(os.path.join(
                args.features_gens_dir_path, 'features{}_to_features{}'.format(input_level_features, input_level_features - 1)),
                map_location='cpu', weights_only=True))
            features_gen.eval()
        else:
            features_gen.init_weights()
        features_gen.to(device)
        parallel.broadcast_module(features_gen)
    discriminator = FeaturesDiscriminator(args.discriminator_norm, dis_type=args.gen_type, dis_level=args.train_block_input - 1)
    discriminator.init_weights()
    discriminator.to(device)
    parallel.broadcast_module(discriminator)

    criterion_discriminator, criterion_generator = get_wgan_losses_fn()
    next_level_features_criterion = L1Loss()
    criterion_features = L1Loss()
    dp = {}
    if world > 1:
        parallel.flatten_grads(features_generators[args.train_block_input - 2]); parallel.flatten_grads(discriminator)
        dp = dict(grad_scale=1.0 / world, zero_grads=True)
    gen_optimizer = optim.Adam(features_generators[args.train_block_input - 2].parameters(), lr=args.lr, betas=(0.5, 0.999), **dp)
    discriminator_optimizer = optim.Adam(discriminator.parameters(), lr=args.lr, betas=(0.5, 0.999), **dp)

    starting_time = time.time()
    iterations = 0
    outputs_dir = os.path.join('features_creation_models', args.model_name)
    temp_results_dir = os.path.join(outputs_dir, 'temp_results')
    models_dir = os.path.join(outputs_dir, 'models_checkpoint')
    writer = None
    if rank == 0:
        for d in (outputs_dir, temp_results_dir, models_dir):
            os.makedirs(d, exist_ok=True)
        writer = imageio.SummaryWriter(os.path.join(outputs_dir, 'summaries'))
    fixed_features = None
    print("Starting Training Loop...")
    features_to_train = args.train_block_input
    for epoch in range(args.epochs):
        for images, _ in train_loader:
            iterations += 1
            if iterations % 30 == 1:
                print('epoch:', epoch, ', iter', iterations, 'start, time =', time.time() - starting_time, 'seconds')
                starting_time = time.time()
            images = imageio.batch_images(images, device)
            with torch.no_grad():
                _, features = classifier(normalizer_clf(images))
            features = list(features)
            if fixed_features is None:
                fixed_features = [torch.clone(f) for f in features]
                if rank == 0:
                    grid = imageio.make_grid(images, padding=2, normalize=False, nrow=8)
                    imageio.save_image(grid, os.path.join(temp_results_dir, 'original_images.jpg'))
            if iterations % (args.discriminator_steps + 1) != 1:
                loss_dict = train_discriminator(features_generators[features_to_train - 2], discriminator, criterion_discriminator,
                                                discriminator_optimizer, features, features_to_train)
                prefix, step = 'D/', iterations
            else:
                loss_dict = train_generator(features_generators[features_to_train - 2], discriminator, classifier, gen_optimizer,
                                            features, features_to_train, criterion_generator, criterion_features,
                                            next_level_features_criterion)
                prefix, step = 'G/f' + str(features_to_train) + '_', iterations // 1 + 1
            for k, v in loss_dict.items():
                if writer is not None:
                    writer.add_scalar(prefix + '%s' % k, float(v), global_step=step)
                if iterations % 30 == 1:
                    print('{}: {:.6f}'.format(k, v))
            if rank == 0 and (iterations < 10000 and iterations % 2000 == 1 or iterations % 4000 == 1):
                for i, features_gen in enumerate(features_generators):
                    features_level = i + 2
                    if features_level == features_to_train:
                        torch.save(features_gen.state_dict(),
                                   models_dir + '/' + args.model_name + '_f{}_to_f{}'.format(features_level, features_level - 1))
                        fake_images = sample(features1_to_image_gen, features_generators, fixed_features, features_level)
                        grid = imageio.make_grid(fake_images, padding=2, normalize=True, nrow=8)
                        imageio.save_image(grid, os.path.join(
                            temp_results_dir, 'res_iter_{}_origin_f{}.jpg'.format(iterations // 2000, features_level)))
            if rank == 0 and iterations % 20000 == 1:
                for i, features_gen in enumerate(features_generators):
                    features_level = i + 2
                    torch.save(features_gen.state_dict(), models_dir + '/' + args.model_name +
                               '_f{}_to_f{}_'.format(features_level, features_level - 1) + str(iterations // 20000))


def build_parser():
    parser = argparse.ArgumentParser('choose hyperparameters')
    parser.add_argument('--batch-size', default=64, type=int)
    parser.add_argument('--lr', default=0.0002, type=float)
    parser.add_argument('--epochs', default=500, type=int)
    parser.add_argument('--model-name')
    parser.add_argument('--discriminator-norm', default='instance_norm', choices=['batch_norm', 'instance_norm', 'layer_norm'])
    parser.add_argument('--gradient-penalty-weight', type=float, default=10.0)
    parser.add_argument('--discriminator-steps', type=int, default=5)
    parser.add_argument('--gen-type', default='res', choices=['default', 'res'])
    parser.add_argument('--train-path', default='/home/dcor/datasets/places365')
    parser.add_argument('--classifier-path', default='/home/dcor/ronmokady/workshop21/team1/try/classifier18')
    parser.add_argument('--features-gens-dir-path', help='the dir that the pre trained generators are located at', required=True)
    # the reference declares int choices without type=int, so only the default (3) is reachable from its CLI (quirk Q6);
    # type=int here makes 2 and 4 reachable too while keeping the default
    parser.add_argument('--train-block-input', default=3, type=int, choices=[2, 3, 4],
                        help='train the block that creates features from the given level')
    return parser


def main(argv=None):
    global args
    args = build_parser().parse_args(argv)
    print(args)
    if args.model_name is None:
        print("Must specify model name")
        exit(1)
    out_dir = './features_creation_models/' + args.model_name
    if parallel.rank_from_env() == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(out_dir + '/hyperparams.txt', 'w') as file:
            for key, value in vars(args).items():
                file.write(key + ': {}\n'.format(value))
    _main()


if __name__ == '__main__':
    main()
