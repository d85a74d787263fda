"""`adp_qtz_fan` flavour: learned levels, fan-in initialisation, 32 levels (adp_qtz_fan.py).

Same public names and constructor signatures as the reference script: MyConv2d, MyLinear,
Residual, PreResidual, ResNet, model_loss and the module-level sharpening factor `beta`.
"""
import torch.nn as nn

from . import layers
from .layers import model_loss  # noqa: F401  (adp_qtz_fan.py:737)

beta = 1  # sharpening factor, read by the layers at call time (adp_qtz_fan.py:160,230)


class _Bound:
    variant = "fan"

    def _global_beta(self):
        return beta


class MyConv2d(_Bound, layers.QConv2d):
    def __init__(self, config, in_channel, out_channel, ksize, stride, padding, bias=False):
        super().__init__()
        self._setup(config, None, in_channel, out_channel, ksize, stride, padding)


class MyLinear(_Bound, layers.QLinear):
    def __init__(self, config, in_feature, out_feature, bias=True):
        super().__init__()
        self._setup(config, None, None, in_feature, out_feature)


class Residual(layers._Block):
    def __init__(self, config, in_channel, increase_dim):
        super().__init__()
        out_channel, stride = (in_channel * 2, 2) if increase_dim else (in_channel, 1)
        self.bn1 = nn.BatchNorm2d(in_channel)
        self.conv1 = MyConv2d(config, in_channel, out_channel, ksize=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.conv2 = MyConv2d(config, out_channel, out_channel, ksize=3, stride=1, padding=1, bias=False)
        self.shortcut = self._shortcut(in_channel, increase_dim)

    def forward(self, x):
        return self._residual_forward(x)


class PreResidual(layers._Block):
    def __init__(self, config, in_channel, increase_dim=False):
        super().__init__()
        self.conv1 = MyConv2d(config, in_channel, in_channel, ksize=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(in_channel)
        self.conv2 = MyConv2d(config, in_channel, in_channel, ksize=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return self._preresidual_forward(x)


class ResNet(layers.QResNet):
    def __init__(self, config, n):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels=3, out_channels=16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn0 = nn.BatchNorm2d(16)
        self.layer1 = self._make_layer(config, n, in_plane=16, first=True)
        self.layer2 = self._make_layer(config, n, in_plane=16, first=False)
        self.layer3 = self._make_layer(config, n, in_plane=32, first=False)
        self.bnlast = nn.BatchNorm2d(64)
        self.ap = nn.AvgPool2d(8)
        self.linear = MyLinear(config, in_feature=64, out_feature=10, bias=True)

    def _make_layer(self, config, n, in_plane, first):
        if first:
            blocks = [PreResidual(config, in_channel=in_plane, increase_dim=False)]
        else:
            blocks = [Residual(config, in_channel=in_plane, increase_dim=True)]
            in_plane *= 2
        blocks += [Residual(config, in_channel=in_plane, increase_dim=False) for _ in range(1, n)]
        return nn.Sequential(*blocks)
