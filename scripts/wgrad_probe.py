"""One weight-gradient call per shape, for rocprofv3 --kernel-trace --stats (development aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantized_cnn_amd.ops as ops  # noqa: E402

B = 32
for cin, cout, ks, s, p, H in [(128, 128, 3, 1, 1, 28), (64, 64, 3, 1, 1, 56), (512, 512, 3, 1, 1, 7)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    y = ops.conv_forward(x, torch.randn(cout, cin, ks, ks, device="cuda"), s, pad=p)
    dy = torch.randn_like(y)
    for _ in range(3):
        ops.conv_wgrad(x, dy, s, ks=ks, pad=p)
torch.cuda.synchronize()
