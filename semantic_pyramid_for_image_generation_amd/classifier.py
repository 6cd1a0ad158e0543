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

    def forward(self, inputs, taps_upto=None):
        """(logits, [relu, layer1, layer2, layer3, layer4, avgpool]) as the reference's Classifier (classifier.py:32-44).
        `taps_upto` (0..4, optional): the caller consumes no tap deeper than `layer<taps_upto>` (train_model.live_taps: the
        reference multiplies every deeper tap by an all-zero mask, quirk Q2) -- the layers behind it are not run and their
        taps, the pooled vector and the logits come back as None.  0: nothing is consumed, nothing runs."""
        if taps_upto is not None:
            if not 0 <= taps_upto <= 4:
                raise ValueError("taps_upto must be in 0..4")
            taps = [None] * 6
            if taps_upto >= 1:
                x = ops.max_pool3s2(run_conv_norm(self.conv1, self.bn1, inputs, ops.ACT_RELU))
                for level in range(1, taps_upto + 1):
                    x = getattr(self, f'layer{level}')(x)
                    taps[level] = x
            return None, taps
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
