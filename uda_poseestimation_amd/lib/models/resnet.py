"""ResNet trunk without the fully connected layer in forward (API mirror of the reference's lib/models/resnet.py:18-62).

The reference subclasses torchvision.models.ResNet; torchvision is not a dependency here.  This module keeps the
same module tree / parameter names (so reference checkpoints load and OldWeightEMA's positional zip matches) but its
sub-modules are PARAMETER CONTAINERS only: the arithmetic of the whole PoseResNet runs in the MI355X executor
(csrc/net.hip) called from PoseResNet.forward.  Calling the trunk on its own is not part of the hot path.
"""
import copy

import torch
import torch.nn as nn

__all__ = ['ResNet', 'Bottleneck', 'resnet50', 'resnet101']


class Bottleneck(nn.Module):
    """torchvision-v1.5 bottleneck (stride on the 3x3), parameters only."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("Bottleneck is a parameter container; run the network through PoseResNet.forward (MI355X executor)")


class ResNet(nn.Module):
    """ResNets without fully connected layer (lib/models/resnet.py:18-49): fc parameters are kept, unused in forward."""

    def __init__(self, block, layers, num_classes=1000, **kwargs):
        super().__init__()
        if block is not Bottleneck:
            raise NotImplementedError("only Bottleneck trunks (resnet50/101) are on the MI355X path")
        self.layers_cfg = list(layers)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._out_features = self.fc.in_features

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*mods)

    def forward(self, x):
        raise RuntimeError("the trunk runs inside PoseResNet.forward on the MI355X executor; it is not callable on its own")

    @property
    def out_features(self) -> int:
        return self._out_features

    def copy_head(self) -> nn.Module:
        return copy.deepcopy(self.fc)


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    """lib/models/resnet.py:52-62.  ImageNet weights cannot be downloaded here (no network): `pretrained` accepts a path
    to a torchvision-format state_dict file, or True/False (True only warns: random init is kept)."""
    model = ResNet(block, layers, **kwargs)
    if isinstance(pretrained, str):
        sd = torch.load(pretrained, map_location='cpu')
        own = model.state_dict()
        model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    elif pretrained:
        import warnings
        warnings.warn(f"pretrained ImageNet weights for {arch} are not available offline; keeping random init "
                      "(pass a state_dict path as pretrained_backbone to load one)")
    return model


def resnet50(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet101(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)
