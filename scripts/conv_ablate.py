"""Development aid: time conv forward (graph of 20 launches) under the experiment build's AQ_CONV_DBG switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
N = 128
def timeg(fn):
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream()); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20): fn()
    torch.cuda.current_stream().wait_stream(st)
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3
for (cin, cout, s, H) in [(16, 16, 1, 32), (32, 32, 1, 16), (64, 64, 1, 8)]:
    x = torch.randn(N, cin, H, H, device="cuda"); W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    y = ops.conv_forward(x, W, s); dy = torch.randn_like(y)
    out = []
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        os.environ["AQ_CONV_DBG"] = str(dbg)
        out.append("dbg%d: %.1f" % (dbg, timeg(lambda: ops.conv_forward(x, W, s))))
    print("%d->%d %dx%d fwd us  " % (cin, cout, H, H), "  ".join(out), flush=True)
