"""Functional torch-CPU restatement of the reference networks, driven by a state_dict-like mapping
`sd` (name -> tensor).  Training-mode BatchNorm updates `sd`'s running buffers in place, like the
reference modules do."""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# generator.py
# --------------------------------------------------------------------------------------------
def _bn(sd, p, x, training):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1) — generator.py:28."""
    y = F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                     training, 0.1, 1e-5)
    if training and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return y


def g_conv_block(sd, p, x, training, deconv, stride, padding, relu):
    """ConvBlock: (de)conv(bias=False) -> BN -> ReLU|Identity — generator.py:15-36."""
    w = sd[p + '.block.0.weight']
    x = F.conv_transpose2d(x, w, None, stride, padding) if deconv else F.conv2d(x, w, None, stride, padding)
    x = _bn(sd, p + '.block.1', x, training)
    return F.relu(x) if relu else x


def g_residual_block(sd, p, x, training, upsample=True):
    """ResidualBlock — generator.py:39-63."""
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    res = g_conv_block(sd, p + '.skip_block', x, training, False, 1, 0, False)
    h = g_conv_block(sd, p + '.conv_block1', x, training, False, 1, 1, True)
    h = g_conv_block(sd, p + '.conv_block2', h, training, False, 1, 1, False)
    return F.relu(h + res)


def g_generator_block(sd, p, gen, feats, training, gen_type, with_features=True):
    """GeneratorBlock — generator.py:66-94."""
    if gen_type == 'default':
        gen = g_conv_block(sd, p + '.conv_block1', gen, training, True, 2, 1, True)
        feats = g_conv_block(sd, p + '.features_conv_block', feats, training, True, 2, 1, True)
        return gen + feats
    gen = g_residual_block(sd, p + '.conv_block1', gen, training)
    if not with_features:
        return gen
    feats = F.interpolate(feats, scale_factor=2, mode='nearest')
    feats = g_conv_block(sd, p + '.features_conv_block', feats, training, False, 1, 1, True)
    return gen + feats


def generator_forward(sd, gen_type, noise, features, masks, training):
    """Generator.forward — generator.py:122-134."""
    x = g_conv_block(sd, 'conv_block1', noise, training, True, 1, 0, True)
    x = g_conv_block(sd, 'conv_block2', x, training, True, 2, 1, True)
    x = g_generator_block(sd, 'conv_block3', x, features[4] * masks[4], training, gen_type)
    x = g_generator_block(sd, 'conv_block4', x, features[3] * masks[3], training, gen_type)
    x = g_generator_block(sd, 'conv_block5', x, features[2] * masks[2], training, gen_type)
    x = g_generator_block(sd, 'conv_block6', x, features[1] * masks[1], training, gen_type)
    if gen_type == 'res':
        x = g_residual_block(sd, 'conv_block7', x, training)
        x = F.conv2d(x, sd['last_conv.weight'], sd['last_conv.bias'], 1, 1)
    else:
        x = F.conv_transpose2d(x, sd['last_conv.weight'], sd['last_conv.bias'], 2, 1)
    return torch.tanh(x)


# --------------------------------------------------------------------------------------------
# discriminator.py
# --------------------------------------------------------------------------------------------
def _d_norm(sd, p, x, norm, training):
    """get_norm_layer — discriminator.py:16-26."""
    if norm == 'none':
        return x
    if norm == 'batch_norm':
        return _bn(sd, p, x, training)
    if norm == 'instance_norm':        # InstanceNorm2d(affine=True): instance statistics in train AND eval
        return F.instance_norm(x, None, None, sd[p + '.weight'], sd[p + '.bias'], True, 0.1, 1e-5)
    if norm == 'layer_norm':           # GroupNorm(1, C)
        return F.group_norm(x, 1, sd[p + '.weight'], sd[p + '.bias'], 1e-5)
    raise NotImplementedError(norm)


def d_conv_block(sd, p, x, norm, training, stride, padding, lrelu):
    """ConvBlock: conv(bias) -> norm -> LeakyReLU(0.2)|Identity — discriminator.py:29-46."""
    x = F.conv2d(x, sd[p + '.block.0.weight'], sd[p + '.block.0.bias'], stride, padding)
    x = _d_norm(sd, p + '.block.1', x, norm, training)
    return F.leaky_relu(x, 0.2) if lrelu else x


def d_residual_block(sd, p, x, norm, training):
    """ResidualBlock — discriminator.py:49-71."""
    res = d_conv_block(sd, p + '.skip_block', x, norm, training, 1, 0, False)
    h = d_conv_block(sd, p + '.conv_block1', x, norm, training, 1, 1, True)
    h = d_conv_block(sd, p + '.conv_block2', h, norm, training, 1, 1, False)
    return F.avg_pool2d(F.leaky_relu(h + res, 0.2), 2)


def d_block(sd, p, x, norm, training, dis_type):
    """DiscriminatorBlock — discriminator.py:74-89."""
    if dis_type == 'default':
        return d_conv_block(sd, p + '.conv_block1', x, norm, training, 2, 1, True)
    return d_residual_block(sd, p + '.conv_block1', x, norm, training)


def discriminator_forward(sd, norm, dis_type, x, training=True):
    """Discriminator.forward — discriminator.py:117-128.  The five blocks are always the 'default'
    (4x4 stride-2) kind: the reference builds them without dis_type (discriminator.py:105-113)."""
    if dis_type == 'default':
        x = F.leaky_relu(F.conv2d(x, sd['first_conv.weight'], sd['first_conv.bias'], 2, 1), 0.2)
    else:
        x = F.leaky_relu(F.conv2d(x, sd['first_conv.weight'], sd['first_conv.bias'], 1, 1), 0.2)
        x = F.avg_pool2d(x, 2)
    for i in range(1, 6):
        x = d_block(sd, 'conv_block%d' % i, x, norm, training, 'default')
    return F.conv2d(x, sd['last_conv.weight'], sd['last_conv.bias'], 1, 0)


# --------------------------------------------------------------------------------------------
# featuresCreatorGen.py
# --------------------------------------------------------------------------------------------
def features1_to_image_forward(sd, f1, training=False):
    """Features1ToImage.forward — featuresCreatorGen.py:16-21."""
    x = g_generator_block(sd, 'conv_block6', f1, None, training, 'res', with_features=False)
    x = g_residual_block(sd, 'conv_block7', x, training)
    return torch.tanh(F.conv2d(x, sd['last_conv.weight'], sd['last_conv.bias'], 1, 1))


def level_up_forward(sd, feats, training=True):
    """LevelUpFeaturesGenerator.forward — featuresCreatorGen.py:41-44."""
    x = g_residual_block(sd, 'res_block1', feats, training, upsample=True)
    return g_residual_block(sd, 'res_block2', x, training, upsample=False)


def features_discriminator_forward(sd, norm, dis_type, dis_level, x, training=True):
    """FeaturesDiscriminator.forward — featuresCreatorGen.py:69-77."""
    if dis_level == 1:
        x = d_block(sd, 'conv_block2', x, norm, training, dis_type)
    if dis_level < 3:
        x = d_block(sd, 'conv_block3', x, norm, training, dis_type)
    x = d_block(sd, 'conv_block4', x, norm, training, dis_type)
    x = d_block(sd, 'conv_block5', x, norm, training, dis_type)
    return F.conv2d(x, sd['last_conv.weight'], sd['last_conv.bias'], 1, 0)


# --------------------------------------------------------------------------------------------
# classifier.py (+ torchvision ResNet-18, restated from its published architecture; eval mode)
# --------------------------------------------------------------------------------------------
def _bn_eval(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.1, 1e-5)


def basic_block(sd, p, x, stride):
    idt = x
    if (p + '.downsample.0.weight') in sd:
        idt = _bn_eval(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0))
    out = F.relu(_bn_eval(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1)))
    out = _bn_eval(sd, p + '.bn2', F.conv2d(out, sd[p + '.conv2.weight'], None, 1, 1))
    return F.relu(out + idt)


def resnet_stage(sd, name, x, stride):
    x = basic_block(sd, name + '.0', x, stride)
    return basic_block(sd, name + '.1', x, 1)


def classifier_forward(sd, inputs):
    """Classifier.forward — classifier.py:32-44 -> (logits, [relu, layer1..4, avgpool])."""
    relu = F.relu(_bn_eval(sd, 'bn1', F.conv2d(inputs, sd['conv1.weight'], None, 2, 3)))
    x = F.max_pool2d(relu, 3, 2, 1)
    l1 = resnet_stage(sd, 'layer1', x, 1)
    l2 = resnet_stage(sd, 'layer2', l1, 2)
    l3 = resnet_stage(sd, 'layer3', l2, 2)
    l4 = resnet_stage(sd, 'layer4', l3, 2)
    ap = F.adaptive_avg_pool2d(l4, (1, 1))
    logits = F.linear(ap.flatten(1), sd['fc.weight'], sd['fc.bias'])
    return logits, [relu, l1, l2, l3, l4, ap]
