"""Feature-creation models — drop-in mirror of the reference `featuresCreatorGen.py`."""
import torch
import torch.nn as nn

from . import ops
from .blocks import init_weights, run_conv
from .discriminator import DiscriminatorBlock
from .generator import GeneratorBlock, ResidualBlock


class Features1ToImage(nn.Module):
    """tap-1 features (B,64,64,64) -> image: the generator's last two stages (reference featuresCreatorGen.py:6-21).
    Loads the shipped `pre trained generators/features1_to_image` state_dict strictly."""

    def __init__(self):
        super().__init__()
        self.conv_block6 = GeneratorBlock(64, 64, 'res', with_features=False)   # 64 -> 128
        self.conv_block7 = ResidualBlock(64, 64)                                # 128 -> 256
        self.last_conv = nn.Conv2d(64, 3, kernel_size=3, stride=1, padding=1)
        self.tanh = nn.Tanh()

    def forward(self, features1):
        x = self.conv_block6(features1, None)
        x = self.conv_block7(x)
        return ops.act(run_conv(self.last_conv, x), ops.ACT_TANH)


def copy_Features1ToImage_from_gen(generator, output_path):
    """Save the tail of a trained res-generator as a Features1ToImage checkpoint (reference :24-29)."""
    f = Features1ToImage()
    f.conv_block6 = generator.conv_block6
    f.conv_block7 = generator.conv_block7
    f.last_conv = generator.last_conv
    torch.save(f.state_dict(), output_path)


class LevelUpFeaturesGenerator(nn.Module):
    """level-L features -> level-(L-1) features: ResidualBlock(up) + ResidualBlock(no up) (reference :32-48)."""

    def __init__(self, input_level_features):
        super().__init__()
        level_to_filters = {1: 64, 2: 128, 3: 256, 4: 512}
        cin = level_to_filters[input_level_features]
        cout = level_to_filters[input_level_features - 1]
        self.res_block1 = ResidualBlock(cin, cin, need_upsample=True)
        self.res_block2 = ResidualBlock(cin, cout, need_upsample=False)

    def forward(self, features2):
        return self.res_block2(self.res_block1(features2))

    def init_weights(self):
        self.res_block1.init_weights()
        self.res_block2.init_weights()


class FeaturesDiscriminator(nn.Module):
    """Critic on feature tensors: the tail of the image critic, starting at `dis_level` (reference :51-86).
    Here `dis_type` IS forwarded to the blocks, so 'res' gives the residual/avg-pool blocks."""

    def __init__(self, norm, dis_type, dis_level):
        super().__init__()
        self.dis_type = dis_type
        self.dis_level = dis_level
        if dis_level == 1:
            self.conv_block2 = DiscriminatorBlock(64, 128, dis_type=dis_type, norm=norm)     # 64 -> 32
        if dis_level < 3:
            self.conv_block3 = DiscriminatorBlock(128, 256, dis_type=dis_type, norm=norm)    # 32 -> 16
        self.conv_block4 = DiscriminatorBlock(256, 512, dis_type=dis_type, norm=norm)        # 16 -> 8
        self.conv_block5 = DiscriminatorBlock(512, 512, dis_type=dis_type, norm=norm)        # 8 -> 4
        self.last_conv = nn.Conv2d(512, 1, 4, stride=1, padding=0)                           # 4 -> 1

    def _blocks(self):
        blocks = []
        if self.dis_level == 1:
            blocks.append(self.conv_block2)
        if self.dis_level < 3:
            blocks.append(self.conv_block3)
        return blocks + [self.conv_block4, self.conv_block5]

    def forward(self, x):
        for blk in self._blocks():
            x = blk(x)
        return run_conv(self.last_conv, x)

    def init_weights(self):
        for blk in self._blocks():
            blk.init_weights()
        self.last_conv.apply(init_weights)
