"""TEST INFRASTRUCTURE ONLY -- stand-in for `torchvision.models.resnet{18,34,50}`.

The reference's embedders (networks/volumetric_avatar/identity_embedder.py:29, expression_embedder.py:370,
head_pose_regressor.py:14) build their backbones with `torchvision.models.<name>(...)`.  torchvision
(pinned `torchvision==0.9.1+cu111`, environment.yml:432) is a third-party dependency that is absent both from the
reference tree and from this container, so the ResNet *body* cannot be executed from its own source here.

This file restates the published architecture (He et al., "Deep Residual Learning for Image Recognition", 2015, in the
"v1.5" form torchvision ships: the stride of a Bottleneck sits on its 3x3 conv) with torchvision's module names and
registration order, because both matter to the reference:
  * the state_dict keys (`conv1.weight`, `layer2.0.downsample.0.weight`, `fc.weight`, ...) are the checkpoint layout;
  * `utils.replace_conv_to_ws_conv` (networks/volumetric_avatar/utils.py:1061-1096) decides which convs become
    weight-standardised from the *order of the children* (a Conv2d whose previous or second-previous sibling is a
    GroupNorm), and `replace_bn_to_gn` (:1020-1038) walks `named_children()`.

oracle/ref_harness.py installs it as `torchvision.models`, so that the reference's own `IdtEmbed`, `ExpressionEmbed`,
`ResNetWrapper` and `HeadPoseRegressor` classes run unmodified on top of it.

PARITY STATUS: wrapper logic (normalisation, alignment warp, pooling order, BN->GN, SN, WS) is pinned to the reference's
classes; the ResNet body itself is pinned only to this restatement of the published architecture ("parity unpinned" for
that part -- there is no torchvision here to run).
"""
import torch
from torch import nn


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(pretrained=False, progress=True, **kw):   # `pretrained` weights cannot be downloaded here: ignored
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(pretrained=False, progress=True, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(pretrained=False, progress=True, **kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)
