"""Development aid for counter collection: a few launches of each conv kernel at the ResNet-20 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
N = 128
for (cin, cout, s, H) in [(16, 16, 1, 32), (16, 32, 2, 32), (32, 32, 1, 16), (32, 64, 2, 16), (64, 64, 1, 8)]:
    x = torch.randn(N, cin, H, H, device="cuda"); W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    y = ops.conv_forward(x, W, s); dy = torch.randn_like(y)
    for _ in range(3):
        ops.conv_forward(x, W, s); ops.conv_dgrad(dy, W, x.shape, s); ops.conv_wgrad(x, dy, s)
torch.cuda.synchronize()
