"""Generator family — drop-in mirror of the reference `generator.py` (same class names, constructor
arguments, `forward` signatures, `init_weights()` and state_dict keys), executed on hand-written HIP
kernels.  The nearest-x2 upsample of the residual blocks is fused into the gather of the consuming
convolutions instead of being materialised (reference generator.py:51-52,85)."""
import torch
import torch.nn as nn

import os

from . import ops
from .blocks import (init_weights, run_conv, run_conv_norm, run_conv_norm_act,  # noqa: F401  (init_weights re-exported)
                     run_norm, run_norm_add_norm_act, tail_takes_half_res, fuses_conv_norm, activation_kind, _single)

ZERO_SKIP = os.environ.get("SEMPYR_ZERO_SKIP", "1") != "0"


class ConvBlock(nn.Module):
    """(de)conv(bias=False) -> BatchNorm2d -> ReLU | Identity   (reference generator.py:15-36)."""

    def __init__(self, channels, num_filters, filter_size, activation='relu', conv_type='deconv', stride=1, padding=1):
        super().__init__()
        conv_cls = nn.ConvTranspose2d if conv_type == 'deconv' else nn.Conv2d
        conv_layer = conv_cls(channels, num_filters, kernel_size=filter_size, padding=padding, stride=stride, bias=False)
        self.block = nn.Sequential(conv_layer, nn.BatchNorm2d(num_filters),
                                   nn.ReLU(inplace=True) if activation == 'relu' else nn.Identity())

    def forward(self, inputs, upsample=False):
        return run_conv_norm_act(self.block, inputs, upsample=upsample)

    def init_weights(self):
        self.block.apply(init_weights)


class ResidualBlock(nn.Module):
    """[nearest x2] -> {1x1 skip + BN  ||  3x3+BN+ReLU -> 3x3+BN} -> add -> ReLU (reference generator.py:39-63)."""

    def __init__(self, in_channels, out_channels, need_upsample=True):
        super().__init__()
        self.in_channels, self.out_channels, self.need_upsample = in_channels, out_channels, need_upsample
        if need_upsample:
            self.upsample = nn.UpsamplingNearest2d(scale_factor=2)
        self.conv_block1 = ConvBlock(in_channels, out_channels, 3, conv_type='conv')
        self.conv_block2 = ConvBlock(out_channels, out_channels, 3, conv_type='conv', activation='none')
        self.skip_block = ConvBlock(in_channels, out_channels, 1, stride=1, padding=0, conv_type='conv', activation='none')
        self.activation = nn.ReLU()

    def forward(self, x):
        up = self.need_upsample           # both consumers gather through the x2 upsample; no 4x tensor is written
        if not (self.conv_block2.block[1].training or self.skip_block.block[1].training):
            # eval mode (eval_model.py / `sample`): each BatchNorm folds into its conv, the add + ReLU into the last one
            # (the 1x1 skip conv commutes with the nearest upsample: it runs at the low resolution and the last conv's
            # epilogue reads it through the upsample)
            sk = self.skip_block.block[0]
            half = up and _single(sk.kernel_size) == 1 and _single(sk.stride) == 1 and _single(sk.padding) == 0 and \
                fuses_conv_norm(self.conv_block2.block[0], self.conv_block2.block[1])
            residual = run_conv_norm(sk, self.skip_block.block[1], x, upsample=up and not half)
            h = self.conv_block1(x, upsample=up)
            return run_conv_norm(self.conv_block2.block[0], self.conv_block2.block[1], h, ops.ACT_RELU, res=residual,
                                 res_half=half)
        # the 1x1 skip convolution commutes with the nearest upsample: it runs on the low-resolution map (a quarter of the
        # work and of the tensor), its BatchNorm statistics are those of the upsampled map, and the tail pass reads it
        # through the upsample
        skip = self.skip_block.block[0]
        half = up and _single(skip.kernel_size) == 1 and _single(skip.stride) == 1 and _single(skip.padding) == 0 and \
            tail_takes_half_res(self.conv_block2.block[1], self.skip_block.block[1])
        residual = run_conv(self.skip_block.block[0], x, upsample=up and not half)
        h = self.conv_block1(x, upsample=up)
        h = run_conv(self.conv_block2.block[0], h, stats=self.conv_block2.block[1].training)
        # BatchNorm(conv2) + BatchNorm(skip) -> add -> ReLU in one apply pass (both blocks have no activation of their own)
        return run_norm_add_norm_act(self.conv_block2.block[1], h, self.skip_block.block[1], residual, ops.ACT_RELU,
                                     b_half=half)

    def init_weights(self):
        for blk in (self.conv_block1, self.conv_block2, self.skip_block):
            blk.init_weights()


class GeneratorBlock(nn.Module):
    """generator branch + (masked) classifier-feature branch, summed (reference generator.py:66-94)."""

    def __init__(self, in_channels, out_channels, gen_type='default', with_features=True):
        super().__init__()
        self.gen_type = gen_type
        self.with_features = with_features
        if gen_type == 'default':
            self.conv_block1 = ConvBlock(in_channels, out_channels, 4, stride=2, padding=1)
            self.features_conv_block = ConvBlock(in_channels, out_channels, 4, stride=2, padding=1)
        else:
            self.conv_block1 = ResidualBlock(in_channels, out_channels)
            if with_features:
                self.upsample = nn.UpsamplingNearest2d(scale_factor=2)
                self.features_conv_block = ConvBlock(in_channels, out_channels, 3, conv_type='conv')

    def forward(self, gen, features=None, mask=None):
        """`mask` (optional): the feature mask of this level; the block then applies it itself (generator.py:126-129) and,
        when the mask is tagged all-zero by the mask setters (every level at or above the selected layer, quirk Q2), does
        not convolve the zeros: the convolution's output is zeros, the BatchNorm / ReLU behind it run as always."""
        gen = self.conv_block1(gen)
        if not self.with_features:
            return gen
        block = self.features_conv_block.block
        if mask is not None:
            if (ZERO_SKIP and getattr(mask, '_sempyr_all_zero', False) and block[0].bias is None
                    and isinstance(block[1], nn.BatchNorm2d) and block[1].training):
                # the feature branch has the generator branch's shape (both are summed below); the tap itself is not read
                # and may be absent (Classifier.forward(..., taps_upto))
                kind, slope = activation_kind(block[2])
                bn = block[1]
                if (not torch.is_grad_enabled() and kind in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU) and bn.affine
                        and (not bn.track_running_stats or bn.momentum is not None)):
                    # detached forward (train_model.py:126): BatchNorm of an all-zero map is its bias, one launch
                    state = (bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)) \
                        if bn.track_running_stats else None
                    return ops.bn_zero_input_add(gen, bn.bias, kind, slope, state)
                y = ops.zero_conv_output(block[0].weight, (gen.shape[0], block[0].out_channels, gen.shape[2], gen.shape[3]))
                return ops.add_act(gen, run_norm(block[1], y, kind, slope))
            if features is None:
                raise ValueError("GeneratorBlock: the classifier tap of this level was not computed but its mask is not all-zero")
            features = ops.mul(features, mask)
        features = self.features_conv_block(features, upsample=(self.gen_type == 'res'))
        return ops.add_act(gen, features)

    def init_weights(self):
        self.conv_block1.init_weights()
        if self.with_features:
            self.features_conv_block.init_weights()


class Generator(nn.Module):
    """z (B,128,1,1) + masked classifier taps 1..4 -> image (B,3,256,256) in [-1,1]
    (reference generator.py:97-145)."""

    def __init__(self, gen_type):
        super().__init__()
        self.gen_type = gen_type
        self.conv_block1 = ConvBlock(128, 512, 4, stride=1, padding=0)      # 1x1 -> 4x4
        self.conv_block2 = ConvBlock(512, 512, 4, stride=2, padding=1)      # 4x4 -> 8x8
        self.conv_block3 = GeneratorBlock(512, 256, gen_type)               # -> 16x16
        self.conv_block4 = GeneratorBlock(256, 128, gen_type)               # -> 32x32
        self.conv_block5 = GeneratorBlock(128, 64, gen_type)                # -> 64x64
        self.conv_block6 = GeneratorBlock(64, 64, gen_type)                 # -> 128x128
        if gen_type == 'default':
            self.last_conv = nn.ConvTranspose2d(64, 3, kernel_size=4, stride=2, padding=1)   # -> 256x256
        else:
            self.conv_block7 = ResidualBlock(64, 64)                        # -> 256x256
            self.last_conv = nn.Conv2d(64, 3, kernel_size=3, stride=1, padding=1)
        self.tanh = nn.Tanh()

    def forward(self, noise, features, masks):
        x = self.conv_block1(noise)
        x = self.conv_block2(x)
        for level, block in ((4, self.conv_block3), (3, self.conv_block4), (2, self.conv_block5), (1, self.conv_block6)):
            x = block(x, features[level], masks[level])
        if self.gen_type == 'res':
            x = self.conv_block7(x)
        return run_conv(self.last_conv, x, act_kind=ops.ACT_TANH)

    def init_weights(self):
        blocks = [self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4, self.conv_block5,
                  self.conv_block6]
        if self.gen_type == 'res':
            blocks.append(self.conv_block7)
        for blk in blocks:
            blk.init_weights()
        self.last_conv.apply(init_weights)
