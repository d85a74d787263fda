"""ResNet-18 for 224 x 224 inputs built from this library's quantized layers.  NO REFERENCE MODEL: the reference
trains ResNet-20 on CIFAR-10 only (adp_qtz_fan.py:696-734); this is the standard ImageNet ResNet-18 graph (7x7 / stride 2
stem, 3x3 / stride 2 max pooling, four stages of two basic blocks with 64 / 128 / 256 / 512 channels, 1x1 / stride 2
projections, global average pooling) with the convolutions of the stages taken from

    config.mode_q = "adaptive" : MyConv2d of adp_qtz_fan (soft level assignment, config.num_level_conv levels)
                    "twn" | "ttq" : ternary.TernaryConv2d
                    "inq"      : ternary.INQConv2d (incremental power-of-two quantisation, 5 bits)

The stem stays full precision, as conv0 does in the reference's ResNet-20.  It exists to run the general convolution
kernels (hundreds of channels, 7x7 and 1x1 kernels) in a whole training step; nothing about it can be compared with
reference code.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers, ops, ternary


def _conv(config, module, cin, cout, ks, stride, pad):
    mode_q = getattr(config, "mode_q", "adaptive")
    if mode_q == "adaptive":
        return module.MyConv2d(config, cin, cout, ks, stride, pad)
    return ternary.make_conv(config, cin, cout, ks, stride, pad)


class _MaxPool3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on the library's kernels: a one-byte window position per output instead of torch's 8-byte
    index, and a gather in the backward pass."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y, idx = ops.maxpool3s2_forward(x)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3s2_backward(dy.contiguous(), idx, ctx.shape)


def max_pool_3s2(pool, x):
    """pool(x) for the stem's nn.MaxPool2d; the library's kernels when it is the usual (3, 2, 1) pool on a HIP fp32 tensor."""
    usual = (pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2)) and pool.padding in (1, (1, 1)) and pool.dilation in (1, (1, 1))
             and not pool.ceil_mode and not pool.return_indices)
    if usual and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        return _MaxPool3s2.apply(x)
    return pool(x)


class BasicBlock(nn.Module):
    def __init__(self, config, module, cin, cout, stride):
        super().__init__()
        self.conv1 = _conv(config, module, cin, cout, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = _conv(config, module, cout, cout, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = _conv(config, module, cin, cout, 1, stride, 0)
            self.bn_down = nn.BatchNorm2d(cout)

    def forward(self, x):
        if self.down is None:
            # identity shortcut: conv1 hands its input through, so that the gradient arriving over the shortcut is added
            # in conv1's input-gradient kernel instead of by a separate accumulation launch
            out, sc = self.conv1(x, split=True)
        else:
            out, sc = self.conv1(x), layers.bn_relu(self.bn_down, self.down(x), relu=False)
        out = layers.bn_relu(self.bn1, out)
        # relu(bn2(conv2(out)) + shortcut) as one kernel (and one in the backward pass)
        return layers.bn_add_relu(self.bn2, self.conv2(out), sc)


class ResNet18(nn.Module):
    def __init__(self, config, module=None, num_classes=1000):
        super().__init__()
        if module is None:
            from . import adp_qtz_fan as module
        self.conv0 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn0 = nn.BatchNorm2d(64)
        self.pool0 = nn.MaxPool2d(3, 2, 1)
        chans, blocks, cin = (64, 128, 256, 512), [], 64
        for i, c in enumerate(chans):
            blocks.append(nn.Sequential(BasicBlock(config, module, cin, c, 1 if i == 0 else 2), BasicBlock(config, module, c, c, 1)))
            cin = c
        self.layer1, self.layer2, self.layer3, self.layer4 = blocks
        self.fc = nn.Linear(512, num_classes)

    def quant_layers(self):
        return [m for m in self.modules() if isinstance(m, layers._QuantBase)]

    def _synthesize_all(self):
        """Adaptive layers: one segmented synthesis launch per 24 layers instead of one per layer (as QResNet does)."""
        ls = self.quant_layers()
        if not ls or ls[0].mode != "train":
            return
        beta = ls[0].current_beta()
        params = []
        for l in ls:
            params += [l.p_c, l.q_level]
        holder = {}
        outs = layers._SynthAll.apply(beta, holder, *params)
        for l, W, slot in zip(ls, outs[:-1], holder["slots"]):
            l._shared = (W, slot)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("the quantized ResNet runs on the HIP device only (input is on %s)" % x.device)
        self._synthesize_all()
        ternary.quantize_all([m for m in self.modules() if isinstance(m, ternary._TernBase)])
        ternary.inq_all([m for m in self.modules() if isinstance(m, ternary.INQConv2d)])
        return self.forward_body(x)

    def forward_body(self, x):
        """The graph with every layer's (quantised / effective) weight already handed down: by forward above, or by a
        fused training step (train.FusedTernaryStep)."""
        c0 = self.conv0
        out = layers.conv3x3(x, c0.weight, 2, None, pad=3)
        out = max_pool_3s2(self.pool0, layers.bn_relu(self.bn0, out))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        out = torch.flatten(F.adaptive_avg_pool2d(out, 1), 1)
        return self.fc(out)
