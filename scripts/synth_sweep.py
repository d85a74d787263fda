"""Development aid: time the synthesis kernels of one library build for several K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

L = 32
rows = 250_000_000 // L
p = torch.rand(rows, L, device="cuda"); q = torch.linspace(-0.25, 0.25, L, device="cuda")
dW = torch.randn(rows, device="cuda"); W = [torch.empty(rows, device="cuda")]; dp = [torch.empty_like(p)]
for K in os.environ.get("KS", "16").split(","):
    os.environ["AQ_SYNTH_K"] = K
    t1 = t(lambda: ops.synth_forward([p], [q], 4.0, out_W=W))
    t2 = t(lambda: ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp), 10)
    print("%s K=%s  fwd %.3f ms %.0f GB/s   bwd %.3f ms %.0f GB/s" % (os.path.basename(os.environ.get("AQ_HIP_LIB", "default")), K,
          t1 * 1e3, rows * 132 / t1 / 1e9, t2 * 1e3, rows * 260 / t2 / 1e9), flush=True)
