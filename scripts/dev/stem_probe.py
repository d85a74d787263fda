"""Development aid: the stem kernels alone (7x7 / stride 2 over RGB at 32 x 224 x 224), for counter passes and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quantized_cnn_amd.ops as ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 3, 224, 224, device="cuda")
W = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
y = ops.conv_forward(x, W, 2, pad=3)
dy = torch.randn_like(y)
for _ in range(10):
    ops.conv_forward(x, W, 2, pad=3)
    ops.conv_wgrad(x, dy, 2, ks=7, pad=3)
torch.cuda.synchronize()
for name, fn in (("fwd", lambda: ops.conv_forward(x, W, 2, pad=3)), ("wgrad", lambda: ops.conv_wgrad(x, dy, 2, ks=7, pad=3))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    fl = 2.0 * B * 64 * 147 * 112 * 112
    print("%s %.1f us  %.1f TF (%.2f)" % (name, t * 1e6, fl / t / 1e12, fl / t / 1e12 / 157.3), flush=True)
