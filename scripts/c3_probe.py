"""Development aid: forward / input-gradient kernel time on ResNet-18 3x3 shapes (events around 10 calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
B = int(os.environ.get("B", "32"))
for cin, cout, s, H in [(64, 64, 1, 56), (128, 128, 1, 28), (256, 256, 1, 14), (512, 512, 1, 7), (64, 128, 2, 56)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    y = ops.conv_forward(x, W, s)
    dy = torch.randn_like(y)
    fl = 2.0 * B * cout * cin * 9 * y.shape[2] * y.shape[3]
    res = []
    for fn in (lambda: ops.conv_forward(x, W, s), lambda: ops.conv_dgrad(dy, W, x.shape, s)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10 * 1e-3
        res.append("%6.1f us %5.1f TF" % (t * 1e6, fl / t / 1e12))
    print("%3d->%3d s%d @%2d: fwd %s   dgrad %s" % (cin, cout, s, H, res[0], res[1]), flush=True)
