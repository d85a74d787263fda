"""Development aid: forward / input-gradient time of one convolution layer (HIP events, 30 launches).
usage: conv_time.py cin cout ks stride pad H [batch]"""
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
out = []
for name, fn in (("fwd", lambda: ops.conv_forward(x, W, s, pad=pad)), ("dgrad", lambda: ops.conv_dgrad(dy, W, x.shape, s, pad=pad))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        fn()
    b.record()
    torch.cuda.synchronize()
    out.append("%s %.1f us" % (name, a.elapsed_time(b) / 30 * 1e3))
print(" ".join(sys.argv[1:]), " | ", "  ".join(out), flush=True)
