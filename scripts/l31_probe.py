"""Development aid: the L = 31 synthesis gradient timed on separate allocations and on views of larger buffers,
alternating with L = 32, several rounds (clock state and allocation placement both move these numbers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops


def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


rows = 7812500
q32 = torch.linspace(-0.25, 0.25, 32, device="cuda")
q31 = torch.linspace(-0.25, 0.25, 31, device="cuda")
dW = torch.randn(rows, device="cuda")
P = torch.rand(rows, 32, device="cuda"); DP = torch.empty_like(P)
p31 = P.view(-1)[:rows * 31].view(rows, 31); dp31 = DP.view(-1)[:rows * 31].view(rows, 31)
s31 = torch.rand(rows, 31, device="cuda"); sdp31 = torch.empty_like(s31)
W = torch.empty(rows, device="cuda")
for rnd in range(4):
    t = timed(lambda: ops.synth_backward([P], [q32], 4.0, [dW], 1e-3, out_dp=[DP]))
    print("round %d  L32           %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 260 / t / 1e9), flush=True)
    t = timed(lambda: ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, out_dp=[dp31]))
    print("round %d  L31 views     %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 252 / t / 1e9), flush=True)
    t = timed(lambda: ops.synth_backward([s31], [q31], 4.0, [dW], 1e-3, out_dp=[sdp31]))
    print("round %d  L31 separate  %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 252 / t / 1e9), flush=True)
    t = timed(lambda: ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, need_dq=[False], out_dp=[dp31]))
    print("round %d  L31 views, frozen levels (no dq)  %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 252 / t / 1e9), flush=True)
    t = timed(lambda: ops.synth_forward([p31], [q31], 4.0, out_W=[W]))
    print("round %d  L31 K1        %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 128 / t / 1e9), flush=True)
q = torch.tensor([-(2.0 ** -k) for k in range(2, 18)] + [2.0 ** -k for k in range(17, 1, -1)], device="cuda")
qb = torch.cat([q[:15], torch.zeros(1, device="cuda"), q[17:]]).contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
dWb = torch.randn(rows, device="cuda", generator=g)
for rnd in range(3):
    t = timed(lambda: ops.synth_backward([p31], [qb], 4.0, [dWb], 1e-3, out_dp=[dp31]), 10)
    print("round %d  L31 views, power-of-two levels with a zero  %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 252 / t / 1e9), flush=True)
    t = timed(lambda: ops.synth_backward([p31], [q31], 4.0, [dWb], 1e-3, out_dp=[dp31]), 10)
    print("round %d  L31 views, linspace levels                  %.3f ms %.0f GB/s" % (rnd, t * 1e3, rows * 252 / t / 1e9), flush=True)
