"""Development aid: a few launches of the general convolution kernels on ResNet-18 layer shapes (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
B = 32
which = sys.argv[1] if len(sys.argv) > 1 else "fwd,dgrad,wgrad"
for cin, cout, ks, s, p, H in [(64, 64, 3, 1, 1, 56), (128, 128, 3, 1, 1, 28), (256, 256, 3, 1, 1, 14), (512, 512, 3, 1, 1, 7), (64, 128, 3, 2, 1, 56),
                               (128, 256, 3, 2, 1, 28), (64, 128, 1, 2, 0, 56)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    W = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
    y = ops.conv_forward(x, W, s, pad=p)
    dy = torch.randn_like(y)
    for _ in range(5):
        if "fwd" in which: ops.conv_forward(x, W, s, pad=p)
        if "dgrad" in which: ops.conv_dgrad(dy, W, x.shape, s, pad=p)
        if "wgrad" in which: ops.conv_wgrad(x, dy, s, ks=ks, pad=p)
    torch.cuda.synchronize()
