"""Development aid: one convolution layer alone (forward, input gradient, weight gradient), for counter passes.
usage: conv_probe.py cin cout ks stride pad H [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quantized_cnn_amd.ops as ops

cin, cout, ks, s, pad, H = [int(v) for v in sys.argv[1:7]]
B = int(sys.argv[7]) if len(sys.argv) > 7 else 32
x = torch.randn(B, cin, H, H, device="cuda")
W = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
y = ops.conv_forward(x, W, s, pad=pad)
dy = torch.randn_like(y)
for _ in range(8):
    ops.conv_forward(x, W, s, pad=pad)
    ops.conv_dgrad(dy, W, x.shape, s, pad=pad)
    ops.conv_wgrad(x, dy, s, ks=ks, pad=pad)
torch.cuda.synchronize()
