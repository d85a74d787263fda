"""Quick per-kernel timing on the GPU (development aid; bench.py is the contract benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops

def timeit(f, iters=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3

def main():
    dev = "cuda"
    which = sys.argv[1:] or ["synth", "conv"]
    if "synth" in which:
        for L in (32, 31):
            for nel in (25e6, 250e6, 1e9):
                rows = int(nel) // L
                p = torch.rand(rows, L, device=dev)
                q = torch.linspace(-0.25, 0.25, L, device=dev)
                dW = torch.randn(rows, device=dev)
                W = [torch.empty(rows, device=dev)]
                dp = [torch.empty_like(p)]
                t, tmin = timeit(lambda: ops.synth_forward([p], [q], 4.0, out_W=W))
                print("K1 soft  L=%d rows=%d: %.3f ms (min %.3f)  %.1f GB/s" % (L, rows, t * 1e3, tmin * 1e3, rows * (4 * L + 4) / t / 1e9), flush=True)
                t, tmin = timeit(lambda: ops.synth_forward([p], [q], 4.0, out_W=W, sumsq=True))
                print("K1 soft+sumsq           : %.3f ms (min %.3f)  %.1f GB/s" % (t * 1e3, tmin * 1e3, rows * (4 * L + 4) / t / 1e9), flush=True)
                t, tmin = timeit(lambda: ops.synth_forward([p], [q], 4.0, out_W=W, codes=True))
                print("K1 soft+codes           : %.3f ms (min %.3f)  %.1f GB/s" % (t * 1e3, tmin * 1e3, rows * (4 * L + 9) / t / 1e9), flush=True)
                t, tmin = timeit(lambda: ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp))
                print("K2 bwd                  : %.3f ms (min %.3f)  %.1f GB/s" % (t * 1e3, tmin * 1e3, rows * (8 * L + 4) / t / 1e9), flush=True)
                del p, dW, W, dp
                torch.cuda.empty_cache()
    if "conv" in which:
        N = 128
        for (cin, cout, s, H) in [(16, 16, 1, 32), (16, 32, 2, 32), (32, 32, 1, 16), (32, 64, 2, 16), (64, 64, 1, 8)]:
            x = torch.randn(N, cin, H, H, device=dev)
            W = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
            y = ops.conv_forward(x, W, s)
            dy = torch.randn_like(y)
            fl = 2.0 * N * cout * cin * 9 * y.shape[2] * y.shape[3]
            for name, f in (("fwd  ", lambda: ops.conv_forward(x, W, s)), ("dgrad", lambda: ops.conv_dgrad(dy, W, x.shape, s)),
                            ("wgrad", lambda: ops.conv_wgrad(x, dy, s)),
                            ("torch", lambda: torch.nn.functional.conv2d(x, W, stride=s, padding=1))):
                t, tmin = timeit(f, 30)
                print("conv %s %d->%d s%d %dx%d: %.1f us (min %.1f)  %.1f TFLOP/s" % (name, cin, cout, s, H, H, t * 1e6, tmin * 1e6, fl / t / 1e12), flush=True)

main()
