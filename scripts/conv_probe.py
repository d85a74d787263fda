"""Development aid: where does a one-hot weight land in the forward convolution?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
torch.manual_seed(0)
x = torch.randn(2, 16, 32, 32, device="cuda")
for (co, ci, tap) in [(0, 0, 4), (3, 5, 4), (15, 15, 4), (3, 5, 0), (7, 2, 8)]:
    W = torch.zeros(16, 16, 3, 3, device="cuda"); W.view(16, 16, 9)[co, ci, tap] = 1
    y = ops.conv_forward(x, W, 1)
    nz = [c for c in range(16) if y[:, c].abs().max() > 0]
    ref = torch.nn.functional.conv2d(x, W, padding=1)
    print((co, ci, tap), "nonzero out channels", nz, "max err", float((y - ref).abs().max()))
    for c in nz:
        for cj in range(16):
            for t in range(9):
                W2 = torch.zeros_like(W); W2.view(16, 16, 9)[c, cj, t] = 1
                r2 = torch.nn.functional.conv2d(x, W2, padding=1)
                if (y[:, c] - r2[:, c]).abs().max() < 1e-5:
                    print("    out ch", c, "behaves like weight at (ci, tap) =", (cj, t))
