"""ResNet-18 parameter containers with torchvision-compatible attribute names and state_dict keys
(`conv1.weight`, `layer2.0.downsample.0.weight`, `fc.bias`, ...).

The reference builds its classifier from `torchvision.models.resnet18(num_classes=365)`
(classifier.py:3,56), a third-party, un-vendored dependency with no pinned version; torchvision is not
installed in this environment.  The architecture restated here is the published torchvision ResNet-18:
7x7/2 stem conv (no bias) -> BN -> ReLU -> 3x3/2 max-pool (pad 1) -> 4 stages of 2 BasicBlocks
(64, 128, 256, 512 channels; stages 2-4 start with stride 2 and a 1x1/2 conv + BN shortcut) ->
global average pool -> fc.  Forward execution is in `classifier.py` (HIP kernels); the modules below
only hold parameters/buffers.
"""
import torch.nn as nn

from . import ops
from .blocks import run_conv_norm


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        return run_basic_block(self, x)


def run_basic_block(block, x):
    """conv3x3-BN-ReLU-conv3x3-BN (+ identity | 1x1/2 conv-BN shortcut) -> add -> ReLU on the HIP path.
    Works on any object with torchvision BasicBlock attributes."""
    identity = x
    if block.downsample is not None:
        identity = run_conv_norm(block.downsample[0], block.downsample[1], x)
    out = run_conv_norm(block.conv1, block.bn1, x, ops.ACT_RELU)
    return run_conv_norm(block.conv2, block.bn2, out, ops.ACT_RELU, res=identity)


class HipStage(nn.Sequential):
    """A ResNet stage whose children keep the names '0', '1', ... (state_dict keys) but run on HIP kernels."""

    def forward(self, x):
        for blk in self:
            x = run_basic_block(blk, x)
        return x


class ResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return HipStage(*layers)


def resnet18(num_classes=1000):
    return ResNet((2, 2, 2, 2), num_classes=num_classes)
