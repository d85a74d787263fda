"""Sweep of the synthesis kernels over tensor sizes (one synthetic segment of R rows x L levels, R*L = 25 M ..
1 B fp32 elements of p_c ~ U[0,1), power-of-two levels as the hint / fix variants initialise them for s = 0.3,
beta = 4, dW ~ N(0,1)): achieved GB/s on algorithmic bytes (K1 4L+4, K1 with codes 4L+5, K2 8L+4, K2 with
Adam 24L+4 bytes per row), HIP events around whole calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops


def timed(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def levels(L):
    half = [2.0 ** -(2 + k) for k in range(16)]                     # 0.25 .. 2^-17
    if L == 32:
        v = [-h for h in half] + half[::-1]
    else:                                                           # 31: 15 per sign around an exact zero
        v = [-h for h in half[:15]] + [0.0] + half[:15][::-1]
    return torch.tensor(v, device="cuda", dtype=torch.float32)


print("%-6s %-13s %10s | %-18s %-18s %-18s %-18s" % ("L", "elements", "rows", "K1 soft", "K1 hard + codes", "K2", "K2 + Adam"))
for L in (32, 31):
    for elems in (25_000_000, 100_000_000, 250_000_000, 1_000_000_000):
        rows = elems // L
        g = torch.Generator(device="cuda").manual_seed(0)
        p = torch.rand(rows, L, device="cuda", generator=g)
        q = levels(L)
        dW = torch.randn(rows, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        W = [torch.empty(rows, device="cuda")]
        dp = [torch.empty_like(p)]
        iters = 20 if elems <= 250_000_000 else 6
        t1 = timed(lambda: ops.synth_forward([p], [q], 4.0, out_W=W), iters)
        t1c = timed(lambda: ops.synth_forward([p], [q], 4.0, soft=False, codes=True), iters)
        t2 = timed(lambda: ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp), iters)
        del dp
        m, v, st = [torch.zeros_like(p)], [torch.zeros_like(p)], [torch.zeros((), device="cuda")]
        t3 = timed(lambda: ops.synth_backward_adam([p], [q], 4.0, [dW], m, v, st, 1e-3, wd=1e-3), max(iters // 2, 3))
        f = lambda t, b: "%7.3f ms %5.0f GB/s" % (t * 1e3, rows * b / t / 1e9)
        print("%-6d %-13d %10d | %-18s %-18s %-18s %-18s" % (L, elems, rows, f(t1, 4 * L + 4), f(t1c, 4 * L + 5), f(t2, 8 * L + 4), f(t3, 24 * L + 4)), flush=True)
        del p, dW, W, m, v
        torch.cuda.empty_cache()
