"""`adp_qtz_fix` flavour: fixed power-of-two levels including 0, num_level-1 = 31 levels, `q_level` frozen (adp_qtz_fix.py).

Same public names and constructor signatures as the reference script: MyConv2d, MyLinear,
Residual, PreResidual, ResNet (built from the scales `s` and bias `v` of a trained full-precision
baseline), model_loss and the module-level sharpening factor `beta`.
"""
import torch.nn as nn

from . import layers
from .layers import model_loss, scales_from_baseline  # noqa: F401

beta = 1  # sharpening factor, read by the layers at call time


class _Bound:
    variant = "fix"

    def _global_beta(self):
        return beta


class MyConv2d(_Bound, layers.QConv2d):
    def __init__(self, config, s, in_channel, out_channel, ksize, stride, padding, bias=False):
        super().__init__()
        self._setup(config, s, in_channel, out_channel, ksize, stride, padding)


class MyLinear(_Bound, layers.QLinear):
    def __init__(self, config, s, v, in_feature, out_feature, bias=True):
        super().__init__()
        self._setup(config, s, v, in_feature, out_feature)


class Residual(layers._Block):
    def __init__(self, config, s1, s2, in_channel, increase_dim):
        super().__init__()
        out_channel, stride = (in_channel * 2, 2) if increase_dim else (in_channel, 1)
        self.bn1 = nn.BatchNorm2d(in_channel)
        self.conv1 = MyConv2d(config, s1, in_channel, out_channel, ksize=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.conv2 = MyConv2d(config, s2, out_channel, out_channel, ksize=3, stride=1, padding=1, bias=False)
        self.shortcut = self._shortcut(in_channel, increase_dim)

    def forward(self, x):
        return self._residual_forward(x)


class PreResidual(layers._Block):
    def __init__(self, config, s1, s2, in_channel, increase_dim=False):
        super().__init__()
        self.conv1 = MyConv2d(config, s1, in_channel, in_channel, ksize=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(in_channel)
        self.conv2 = MyConv2d(config, s2, in_channel, in_channel, ksize=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return self._preresidual_forward(x)


class ResNet(layers.QResNet):
    """s: 20 scales (s[0], the full-precision conv0, is unused), v: baseline linear bias (adp_qtz_fix.py:795-820)."""

    def __init__(self, config, s, v):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels=3, out_channels=16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn0 = nn.BatchNorm2d(16)
        it = iter(s[1:19])

        def stage(in_plane, first):
            if first:
                blocks = [PreResidual(config, next(it), next(it), in_channel=in_plane, increase_dim=False)]
            else:
                blocks = [Residual(config, next(it), next(it), in_channel=in_plane, increase_dim=True)]
                in_plane *= 2
            blocks += [Residual(config, next(it), next(it), in_channel=in_plane, increase_dim=False) for _ in range(2)]
            return nn.Sequential(*blocks)

        self.layer1 = stage(16, True)
        self.layer2 = stage(16, False)
        self.layer3 = stage(32, False)
        self.bnlast = nn.BatchNorm2d(64)
        self.ap = nn.AvgPool2d(8)
        self.linear = MyLinear(config, s[19], v, in_feature=64, out_feature=10, bias=True)
