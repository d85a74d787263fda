"""Development aid: convolution forward with and without the statistics epilogue, and the batch norm that
consumes either (graph replay of 50 launches, us per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops

def timed(fn, reps=50, rounds=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


def main():
    N = 128
    for (cin, cout, s, H) in [(3, 16, 1, 32), (16, 16, 1, 32), (16, 32, 2, 32), (32, 32, 1, 16), (32, 64, 2, 16), (64, 64, 1, 8)]:
        x = torch.randn(N, cin, H, H, device="cuda"); W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
        y, st = ops.conv_forward(x, W, s, stats=True)
        g_, b_ = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        a = timed(lambda: ops.conv_forward(x, W, s))
        b = timed(lambda: ops.conv_forward(x, W, s, stats=True))
        c = timed(lambda: ops.bn_relu_forward(y, g_, b_))
        d = timed(lambda: ops.bn_relu_forward(y, g_, b_, conv_stats=st))
        print("%2d->%2d s%d %2dx%2d  conv %.2f  conv+stats %.2f  |  bn (stats+apply) %.2f  bn (conv stats) %.2f  | P=%d" % (cin, cout, s, H, H, a, b, c, d, st[1].P), flush=True)


if __name__ == "__main__":
    main()
