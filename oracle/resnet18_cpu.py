"""A plain torch.nn ResNet-18 (torchvision's published architecture and state_dict key layout) used by
the oracle and by oracle/gen_golden.py as the `model` the reference's Classifier wrapper is given —
torchvision itself is not installed here (SURVEY.md §8c gap 1)."""
import torch.nn as nn


class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNet18(nn.Module):
    def __init__(self, num_classes=365):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_Block(64, 64, 1), _Block(64, 64, 1))
        self.layer2 = nn.Sequential(_Block(64, 128, 2), _Block(128, 128, 1))
        self.layer3 = nn.Sequential(_Block(128, 256, 2), _Block(256, 256, 1))
        self.layer4 = nn.Sequential(_Block(256, 512, 2), _Block(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def randomize_bn_stats(model, gen):
    """Give eval-mode BatchNorms non-trivial statistics/affines so that parity tests exercise them."""
    import torch
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
