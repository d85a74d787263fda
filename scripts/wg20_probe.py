"""Development aid: weight gradient of the ResNet-20 stage-3 shapes, 20 launches in a replayed graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
B = 128
for cin, cout, s, H in [(64, 64, 1, 8), (32, 64, 2, 16), (32, 32, 1, 16)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    Ho = (H + 2 - 3) // s + 1
    dy = torch.randn(B, cout, Ho, Ho, device="cuda")
    out = torch.empty(cout, cin, 3, 3, device="cuda")
    fn = lambda: ops.conv_wgrad(x, dy, s, out=out)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(20): fn()
    torch.cuda.current_stream().wait_stream(st)
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): gr.replay()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 100 * 1e-3
    fl = 2.0 * B * cout * cin * 9 * Ho * Ho
    print("%3d->%3d s%d @%2d N=%d: %6.1f us %5.1f TF" % (cin, cout, s, H, B, t * 1e6, fl / t / 1e12), flush=True)
