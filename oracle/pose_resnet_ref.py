"""Plain ``torch.nn`` CPU restatement of the reference pose network (oracle, test-only).

Follows:
  * lib/models/pose_resnet.py:11-56   Upsampling (3x ConvTranspose2d k4 s2 p1 + BN + ReLU, init N(0,0.001))
  * lib/models/pose_resnet.py:59-91   PoseResNet (backbone -> upsampling -> 1x1 head, init N(0,0.001)/0)
  * lib/models/pose_resnet.py:94-126  factories pose_resnet50 / pose_resnet101 ([3,4,6,3] / [3,4,23,3])
  * lib/models/resnet.py:18-49        ResNet.forward without avgpool/fc (fc parameters are kept)
  * torchvision.models.resnet (NOT in the reference tree; published "ResNet v1.5" algorithm):
    Bottleneck = 1x1 -> BN -> ReLU -> 3x3(stride) -> BN -> ReLU -> 1x1(x4) -> BN -> (+ identity or
    1x1(stride)+BN) -> ReLU; stem 7x7 s2 p3 + BN + ReLU + maxpool 3x3 s2 p1; conv init
    kaiming_normal(fan_out, relu); BN (1, 0); eps 1e-5; momentum 0.1.  parity unpinned (see package doc).
"""
import torch
import torch.nn as nn


class BottleneckRef(nn.Module):
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

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNetTrunkRef(nn.Module):
    """Trunk with torchvision's attribute names so state_dict keys match SURVEY Appendix B."""

    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))   # unused by forward (resnet.py:37-39)
        self.fc = nn.Linear(2048, 1000)               # kept: it is in state_dict and EMA'd
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.out_features = 2048

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        seq = [BottleneckRef(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            seq.append(BottleneckRef(self.inplanes, planes))
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class UpsamplingRef(nn.Sequential):
    def __init__(self, in_channel=2048, hidden=(256, 256, 256), bias=False):
        mods = []
        for h in hidden:
            mods += [nn.ConvTranspose2d(in_channel, h, 4, stride=2, padding=1, output_padding=0, bias=bias),
                     nn.BatchNorm2d(h), nn.ReLU(inplace=True)]
            in_channel = h
        super().__init__(*mods)
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
                if bias:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class PoseResNetRef(nn.Module):
    def __init__(self, layers, num_keypoints, deconv_with_bias=False):
        super().__init__()
        self.backbone = ResNetTrunkRef(layers)
        self.upsampling = UpsamplingRef(2048, bias=deconv_with_bias)
        self.head = nn.Conv2d(256, num_keypoints, 1)
        nn.init.normal_(self.head.weight, std=0.001)
        nn.init.constant_(self.head.bias, 0)

    def forward(self, x):
        return self.head(self.upsampling(self.backbone(x)))


def pose_resnet50_ref(num_keypoints, **kw):
    return PoseResNetRef([3, 4, 6, 3], num_keypoints, **kw)


def pose_resnet101_ref(num_keypoints, **kw):
    return PoseResNetRef([3, 4, 23, 3], num_keypoints, **kw)


def tiny_pose_resnet_ref(num_keypoints, layers=(1, 1, 1, 1)):
    """Same architecture with one bottleneck per stage: for fast parity tests."""
    return PoseResNetRef(list(layers), num_keypoints)
