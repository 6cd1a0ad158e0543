"""Classifier tap wrapper — drop-in mirror of the reference `classifier.py:11-44`: wraps a ResNet-18
(365 Places classes) and returns `(logits, [relu, layer1, layer2, layer3, layer4, avgpool])`.

Every stage runs on the HIP operators; eval-mode BatchNorm is a single fused affine(+residual)+ReLU
pass.  `classifier.layer2/3/4(x)` can be called directly, as `train_feature_creation_gen.py:41-46` does.
"""
import torch.nn as nn

from . import ops
from .blocks import run_conv_norm
from .resnet import HipStage, resnet18  # noqa: F401  (resnet18 re-exported for model construction)


def _as_hip_stage(stage):
    if isinstance(stage, HipStage):
        return stage
    return HipStage(*list(stage.children()))     # e.g. a torchvision nn.Sequential of BasicBlocks: same child names


class Classifier(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.conv1 = model.conv1
        self.bn1 = model.bn1
        self.relu = model.relu
        self.maxpool = model.maxpool
        self.layer1 = _as_hip_stage(model.layer1)
        self.layer2 = _as_hip_stage(model.layer2)
        self.layer3 = _as_hip_stage(model.layer3)
        self.layer4 = _as_hip_stage(model.layer4)
        self.avgpool = model.avgpool
        self.flat = nn.Flatten()
        self.fc = model.fc
        mp = self.maxpool
        geom = tuple(v if isinstance(v, int) else v[0] for v in (mp.kernel_size, mp.stride, mp.padding))
        if geom != (3, 2, 1):
            raise NotImplementedError(f"stem max-pool {geom} has no HIP kernel (ResNet uses 3/2/1)")

    def forward(self, inputs):
        relu = run_conv_norm(self.conv1, self.bn1, inputs, ops.ACT_RELU)
        maxpool = ops.max_pool3s2(relu)
        layer1 = self.layer1(maxpool)
        layer2 = self.layer2(layer1)
        layer3 = self.layer3(layer2)
        layer4 = self.layer4(layer3)
        avgpool = ops.global_avg_pool(layer4)
        flat = avgpool.reshape(avgpool.shape[0], -1)
        outputs = ops.linear(flat, self.fc.weight, self.fc.bias)
        return outputs, [relu, layer1, layer2, layer3, layer4, avgpool]
