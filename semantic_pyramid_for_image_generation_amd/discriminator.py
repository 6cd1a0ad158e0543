"""Discriminator (critic) family — drop-in mirror of the reference `discriminator.py`, executed on
hand-written HIP kernels; every operator here supports the double backward that the WGAN-GP gradient
penalty needs."""
import functools

import torch.nn as nn

from . import ops
from .blocks import activation_kind, init_weights, run_conv, run_conv_norm_act, run_norm  # noqa: F401


def get_norm_layer(norm):
    """'none' | 'batch_norm' | 'instance_norm' | 'layer_norm' -> layer factory (reference discriminator.py:16-26)."""
    if norm == 'none':
        return nn.Identity
    if norm == 'batch_norm':
        return nn.BatchNorm2d
    if norm == 'instance_norm':
        return functools.partial(nn.InstanceNorm2d, affine=True)
    if norm == 'layer_norm':
        return lambda num_features: nn.GroupNorm(1, num_features)
    raise NotImplementedError


class ConvBlock(nn.Module):
    """conv(bias=True) -> norm -> LeakyReLU(0.2) | Identity (reference discriminator.py:29-46)."""

    def __init__(self, channels, num_filters, filter_size, activation='relu', stride=1, padding=1, norm='batch_norm'):
        super().__init__()
        Norm = get_norm_layer(norm)
        self.block = nn.Sequential(nn.Conv2d(channels, num_filters, kernel_size=filter_size, padding=padding, stride=stride),
                                   Norm(num_filters),
                                   nn.LeakyReLU(0.2, inplace=True) if activation == 'relu' else nn.Identity())

    def forward(self, inputs):
        return run_conv_norm_act(self.block, inputs)

    def init_weights(self):
        self.block.apply(init_weights)


class ResidualBlock(nn.Module):
    """{1x1 skip + norm || 3x3+norm+LReLU -> 3x3+norm} -> add -> LReLU -> AvgPool2 (reference discriminator.py:49-71)."""

    def __init__(self, in_channels, out_channels, norm='batch_norm'):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_block1 = ConvBlock(in_channels, in_channels, 3, norm=norm)
        self.conv_block2 = ConvBlock(in_channels, out_channels, 3, norm=norm, activation='none')
        self.skip_block = ConvBlock(in_channels, out_channels, 1, stride=1, padding=0, norm=norm, activation='none')
        self.activation = nn.LeakyReLU(0.2, inplace=True)
        self.downsample = nn.AvgPool2d(2)

    def forward(self, x):
        residual = self.skip_block(x)
        h = self.conv_block2(self.conv_block1(x))
        h = ops.add_act(h, residual, ops.ACT_LRELU, float(self.activation.negative_slope))
        return ops.avg_pool2(h)

    def init_weights(self):
        for blk in (self.conv_block1, self.conv_block2, self.skip_block):
            blk.init_weights()


class DiscriminatorBlock(nn.Module):
    """'default': 4x4 stride-2 ConvBlock ; otherwise the residual block (reference discriminator.py:74-89)."""

    def __init__(self, in_channels, out_channels, dis_type='default', norm='batch_norm'):
        super().__init__()
        self.dis_type = dis_type
        if dis_type == 'default':
            self.conv_block1 = ConvBlock(in_channels, out_channels, 4, stride=2, padding=1, norm=norm)
        else:
            self.conv_block1 = ResidualBlock(in_channels, out_channels, norm)

    def forward(self, dis):
        return self.conv_block1(dis)

    def init_weights(self):
        self.conv_block1.init_weights()


class Discriminator(nn.Module):
    """(B,3,256,256) -> (B,1,1,1) critic (reference discriminator.py:92-137).  As in the reference, the
    five DiscriminatorBlocks are built without `dis_type`, so they are always 4x4 stride-2 convs;
    `dis_type='res'` only changes the stem."""

    def __init__(self, norm, dis_type):
        super().__init__()
        self.dis_type = dis_type
        if dis_type == 'default':
            self.first_conv = nn.Conv2d(3, 64, kernel_size=4, stride=2, padding=1)
            self.leaky_relu = nn.LeakyReLU(0.2, True)
        else:
            self.first_conv = nn.Conv2d(3, 64, kernel_size=3, stride=1, padding=1)
            self.leaky_relu = nn.LeakyReLU(0.2, True)
            self.avg_pool = nn.AvgPool2d(2)
        self.conv_block1 = DiscriminatorBlock(64, 64, norm=norm)      # 128 -> 64
        self.conv_block2 = DiscriminatorBlock(64, 128, norm=norm)     # -> 32
        self.conv_block3 = DiscriminatorBlock(128, 256, norm=norm)    # -> 16
        self.conv_block4 = DiscriminatorBlock(256, 512, norm=norm)    # -> 8
        self.conv_block5 = DiscriminatorBlock(512, 512, norm=norm)    # -> 4
        self.last_conv = nn.Conv2d(512, 1, 4, stride=1, padding=0)    # -> 1

    def forward(self, x):
        x = run_conv(self.first_conv, x, act_kind=ops.ACT_LRELU, slope=float(self.leaky_relu.negative_slope),
                     pool=(self.dis_type == 'res'))
        for blk in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5):
            x = blk(x)
        return run_conv(self.last_conv, x)

    def init_weights(self):
        self.first_conv.apply(init_weights)
        for blk in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5):
            blk.init_weights()
        self.last_conv.apply(init_weights)
