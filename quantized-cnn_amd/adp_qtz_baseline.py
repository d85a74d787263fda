"""Full-precision ResNet-20 of the reference's `adp_qtz_baseline.py` (the model whose trained weights
give the power-of-two variants their scales, adp_qtz_fix.py:200-215), with the same class names,
constructor signatures and state_dict keys, running on this library's kernels: MFMA 3x3 convolutions
(forward / dgrad / wgrad), fused BatchNorm+ReLU, shortcut kernels.  The convolution weights stay
ordinary nn.Conv2d parameters, so reference checkpoints load unchanged and torch optimizers
(optim.SGD(momentum=0.9), adp_qtz_baseline.py:164) apply as they are.
"""
import torch.nn as nn

from . import layers


def _ours(conv, x):
    return (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride in ((1, 1), (2, 2)) and conv.bias is None
            and conv.groups == 1 and conv.dilation == (1, 1) and x.is_cuda)


def _conv(conv, x, res=None, bn=None, split=False):
    """nn.Conv2d module applied through the MFMA kernels when it is a 3x3 / pad 1 / stride 1|2 / no-bias
    convolution on the HIP device (all convolutions of this model are); bn: BatchNorm2d whose relu(bn(x))
    is the input; split: also return x for the caller's shortcut."""
    if _ours(conv, x):
        if bn is not None:
            return layers.bn_relu_conv(bn, x, conv.weight, conv.stride[0], None, res, split, stats=conv.training)
        return layers.conv3x3(x, conv.weight, conv.stride[0], None, res, stats=conv.training)
    xs = x
    if bn is not None:
        x = layers.bn_relu(bn, x)
    y = conv(x)
    y = y if res is None else y + res
    return (y, xs) if split else y


class Residual(layers._Block):
    """adp_qtz_baseline.py:453-481"""

    def __init__(self, in_channel, increase_dim):
        super().__init__()
        out_channel, stride = (in_channel * 2, 2) if increase_dim else (in_channel, 1)
        self.bn1 = nn.BatchNorm2d(in_channel)
        self.conv1 = nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.conv2 = nn.Conv2d(out_channel, out_channel, kernel_size=3, stride=1, padding=1, bias=False)
        self.shortcut = self._shortcut(in_channel, increase_dim)

    def forward(self, x):
        out, xs = _conv(self.conv1, x, bn=self.bn1, split=True)
        return _conv(self.conv2, out, res=self._apply_shortcut(xs), bn=self.bn2)


class PreResidual(layers._Block):
    """adp_qtz_baseline.py:484-497"""

    def __init__(self, in_channel, increase_dim=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channel, in_channel, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(in_channel)
        self.conv2 = nn.Conv2d(in_channel, in_channel, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return _conv(self.conv2, _conv(self.conv1, x), res=x, bn=self.bn2)


class ResNet(nn.Module):
    """adp_qtz_baseline.py:500-539"""

    def __init__(self, n):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels=3, out_channels=16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn0 = nn.BatchNorm2d(16)
        self.layer1 = self._make_layer(n, in_plane=16, first=True)
        self.layer2 = self._make_layer(n, in_plane=16, first=False)
        self.layer3 = self._make_layer(n, in_plane=32, first=False)
        self.bnlast = nn.BatchNorm2d(64)
        self.ap = nn.AvgPool2d(8)
        self.linear = nn.Linear(64, 10)

    def _make_layer(self, n, in_plane, first):
        if first:
            blocks = [PreResidual(in_channel=in_plane, increase_dim=False)]
        else:
            blocks = [Residual(in_channel=in_plane, increase_dim=True)]
            in_plane *= 2
        blocks += [Residual(in_channel=in_plane, increase_dim=False) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("this ResNet runs on the HIP device only (input is on %s)" % x.device)
        out = layers.bn_relu(self.bn0, _conv(self.conv0, x))
        out = self.layer3(self.layer2(self.layer1(out)))
        out = layers.bn_relu_pool(self.bnlast, self.ap, out)
        out = out.view(out.size(0), -1)
        return self.linear(out)


def model_loss(config, model):
    """l2_reg * sum of squares of every parameter whose name contains "weight" (adp_qtz_baseline.py:542-548)."""
    import torch
    ws = [p for n, p in model.named_parameters() if "weight" in n]
    return torch.cat([w.reshape(-1) for w in ws]).pow(2).sum() * config.l2_reg
