"""Seeded input recipes shared by oracle/gen_golden.py (which feeds them to the reference) and by the
tests (which feed them to the oracle and to the HIP path).  Only torch's CPU generator and Python's
`random.Random` are used, so the same tensors are rebuilt anywhere from the seed alone."""
import random

import torch

from . import masks as omasks

CLF_MEAN = (0.485, 0.456, 0.406)
CLF_STD = (0.229, 0.224, 0.225)


def tap_shapes(B, size=256):
    """Shapes of the six classifier taps for a size x size input (classifier.py:33-44)."""
    s = size
    return [(B, 64, s // 2, s // 2), (B, 64, s // 4, s // 4), (B, 128, s // 8, s // 8), (B, 256, s // 16, s // 16),
            (B, 512, s // 32, s // 32), (B, 512, 1, 1)]


def image_batch(seed, B, size=256):
    """What ToTensor() yields after Resize/CenterCrop (train_model.py:233-238): fp32 in [0,1)."""
    g = torch.Generator().manual_seed(1234 + seed)
    return torch.rand(B, 3, size, size, generator=g)


def normalize_clf(x):
    m = torch.tensor(CLF_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    s = torch.tensor(CLF_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - m) / s


def generator_inputs(seed, B=2):
    """z ~ N(0,1), non-negative features of the tap shapes, part-2 masks for layer (seed % 4) + 1."""
    g = torch.Generator().manual_seed(1000 + seed)
    shapes = tap_shapes(B)
    z = torch.randn(B, 128, 1, 1, generator=g)
    feats = [torch.rand(s, generator=g) for s in shapes]
    ms, _ = omasks.masks_part2(shapes, seed % 4 + 1, random.Random(seed))
    return z, feats, [torch.from_numpy(m) for m in ms]


def perturb_bn_buffers(module, seed):
    """Non-trivial running statistics for eval-mode tests (visits BatchNorm2d modules in .modules() order)."""
    g = torch.Generator().manual_seed(2000 + seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.05)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.1)


def make_resnet(seed, num_classes=365):
    from .resnet18_cpu import ResNet18, randomize_bn_stats
    torch.manual_seed(3000 + seed)
    model = ResNet18(num_classes)
    randomize_bn_stats(model, torch.Generator().manual_seed(4000 + seed))
    model.eval()
    return model
