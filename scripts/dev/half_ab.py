"""Development aid: K1 / K2 / K2+Adam at L = 32 with the library's 32-row wavefront forms on or off
(AQ_SYNTH_HALF bit mask, read once per process) and the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import quantized_cnn_amd.ops as ops


def timed(fn, iters, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


print("AQ_SYNTH_HALF =", os.environ.get("AQ_SYNTH_HALF"))
L = 32
half = [2.0 ** -(2 + k) for k in range(16)]
q = torch.tensor([-h for h in half] + half[::-1], device="cuda")
for elems in (50_000_000, 250_000_000, 1_000_000_000):
    rows = elems // L
    p = torch.rand(rows, L, device="cuda")
    dW = torch.randn(rows, device="cuda")
    W = [torch.empty(rows, device="cuda")]
    dp = [torch.empty_like(p)]
    it = 20 if elems <= 250_000_000 else 6
    t1 = timed(lambda: ops.synth_forward([p], [q], 4.0, out_W=W), it)
    t2 = timed(lambda: ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp), it)
    del dp
    m, v, st = [torch.zeros_like(p)], [torch.zeros_like(p)], [torch.zeros((), device="cuda")]
    t3 = timed(lambda: ops.synth_backward_adam([p], [q], 4.0, [dW], m, v, st, 1e-3, wd=1e-3), max(it // 2, 3))
    f = lambda t, b: "%7.3f ms %5.0f GB/s" % (t * 1e3, rows * b / t / 1e9)
    print("%-11d K1 %s | K2 %s | K2+Adam %s" % (elems, f(t1, 132), f(t2, 260), f(t3, 772)), flush=True)
    del p, dW, W, m, v
    torch.cuda.empty_cache()
