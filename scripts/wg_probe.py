"""Development aid: weight-gradient kernel time on ResNet-18 shapes (events around 10 calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
B = 32
for cin, cout, s, H in [(64, 64, 1, 56), (128, 128, 1, 28), (256, 256, 1, 14), (512, 512, 1, 7), (64, 128, 2, 56)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    Ho = (H + 2 - 3) // s + 1
    dy = torch.randn(B, cout, Ho, Ho, device="cuda")
    out = torch.empty(cout, cin, 3, 3, device="cuda")
    for _ in range(3): ops.conv_wgrad(x, dy, s, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.conv_wgrad(x, dy, s, out=out)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10 * 1e-3
    fl = 2.0 * B * cout * cin * 9 * Ho * Ho
    print("%3d->%3d s%d @%2d: %6.1f us %5.1f TF" % (cin, cout, s, H, t * 1e6, fl / t / 1e12), flush=True)
