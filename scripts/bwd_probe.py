"""Development aid: input gradient + weight-gradient slabs as two launches against aq_conv_bwd's one launch
(graph replay of 50 pairs, us per pair)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
from stats_probe import timed

N = 128
for (cin, cout, s, H) in [(16, 16, 1, 32), (16, 32, 2, 32), (32, 32, 1, 16), (32, 64, 2, 16), (64, 64, 1, 8)]:
    xb = torch.randn(N, cin, H, H, device="cuda"); W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    g_, b_ = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    z, mean, rstd = ops.bn_relu_forward(xb, g_, b_)
    dy = torch.randn_like(ops.conv_forward(z, W, s))
    bn = (xb, g_, b_, mean, rstd)
    def two():
        ops.conv_wgrad_partial(z, dy, s)
        ops.conv_dgrad(dy, W, z.shape, s, bn=bn)
    a = timed(two)
    b = timed(lambda: ops.conv_backward(dy, W, z, s, bn=bn))
    c = timed(lambda: ops.conv_wgrad_partial(z, dy, s))
    d = timed(lambda: ops.conv_dgrad(dy, W, z.shape, s, bn=bn))
    print("%2d->%2d s%d %2dx%2d  two launches %.2f (wgrad %.2f + dgrad %.2f)   aq_conv_bwd %.2f" % (cin, cout, s, H, H, a, c, d, b), flush=True)
