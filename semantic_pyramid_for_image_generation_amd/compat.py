"""Loading the reference's classifier checkpoint.

`train_model.py:250` does `torch.load(args.classifier_path)` on a *whole-module* pickle of
`classifier.Classifier` wrapping `torchvision.models.resnet.ResNet`.  Here the file is read with
`weights_only=True` plus an explicit allow-list that maps those pickled class names onto this
package's parameter containers, so nothing from the file is executed and neither the reference's
`classifier.py` nor torchvision has to be importable.  A plain state_dict (of the Classifier or of a
ResNet-18) is accepted as well.
"""
import collections

import torch
import torch.nn as nn

from . import classifier as _clf
from . import resnet as _resnet


def _allow_list():
    pairs = [
        (_clf.Classifier, 'classifier.Classifier'),
        (_resnet.ResNet, 'torchvision.models.resnet.ResNet'),
        (_resnet.BasicBlock, 'torchvision.models.resnet.BasicBlock'),
    ]
    plain = [nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d, nn.Sequential, nn.AdaptiveAvgPool2d, nn.Linear, nn.Flatten,
             nn.Identity, collections.OrderedDict, set, _clf.Classifier, _resnet.ResNet, _resnet.BasicBlock, _resnet.HipStage]
    return pairs + plain


def _rewrap(obj):
    """Objects rebuilt from the pickle bypass __init__: re-create the wrapper so stages become HipStage."""
    if isinstance(obj, _clf.Classifier):
        shell = nn.Module()
        for name in ('conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4', 'avgpool', 'fc'):
            setattr(shell, name, getattr(obj, name))
        return _clf.Classifier(shell)
    if isinstance(obj, _resnet.ResNet):
        return _clf.Classifier(obj)
    return None


def load_classifier(path, device=None):
    with torch.serialization.safe_globals(_allow_list()):
        obj = torch.load(path, map_location='cpu', weights_only=True)
    model = _rewrap(obj) if isinstance(obj, nn.Module) else None
    if model is None:
        if isinstance(obj, dict) and 'state_dict' in obj:            # e.g. resnet18_places365.pth.tar (classifier.py:57-59)
            obj = {k.replace('module.', ''): v for k, v in obj['state_dict'].items()}
        if not isinstance(obj, dict):
            raise TypeError(f"{path}: expected a pickled Classifier/ResNet module or a state_dict, got {type(obj).__name__}")
        num_classes = obj['fc.weight'].shape[0]
        model = _clf.Classifier(_resnet.resnet18(num_classes))
        model.load_state_dict(obj, strict=True)
    if device is not None:
        model.to(device)
    return model
