"""Development aid for counter collection: a few launches of the synthesis forward / backward
kernels on the sweep tensor and nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops
L = 32
rows = 250_000_000 // L
p = torch.rand(rows, L, device="cuda"); q = torch.linspace(-0.25, 0.25, L, device="cuda")
dW = torch.randn(rows, device="cuda"); W = [torch.empty(rows, device="cuda")]; dp = [torch.empty_like(p)]
for _ in range(5):
    ops.synth_forward([p], [q], 4.0, out_W=W)
for _ in range(5):
    ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp)
del dp
m1, m2, stp = torch.zeros_like(p), torch.zeros_like(p), torch.zeros((), device="cuda")
for _ in range(5):
    ops.synth_backward_adam([p], [q], 4.0, [dW], [m1], [m2], [stp], 1e-2, wd=1e-3)
torch.cuda.synchronize()
print("rows", rows, "alg bytes fwd", rows * 132, "bwd", rows * 260, "bwd+adam", rows * 772)
del m1, m2
for _ in range(5):
    ops.synth_forward([p], [q], 4.0, soft=False, codes=True)
w = p.view(-1)
for _ in range(5):
    ops.tern_forward([w], "twn", decoded=False)
print("codes-only alg bytes", rows * 133, "ternary quantise alg bytes", w.numel() * 9)
torch.cuda.synchronize()
# 31 levels (adp_qtz_fix: 124-byte rows): forward, gradient without dq (frozen levels) and with dq; exact forward with W + codes
p31 = p.view(-1)[:rows * 31].view(rows, 31)
q31 = torch.cat([q[:15], torch.zeros(1, device="cuda"), q[17:]]).contiguous()
dp31 = [torch.empty_like(p).view(-1)[:rows * 31].view(rows, 31)]
for _ in range(5):
    ops.synth_forward([p31], [q31], 4.0, out_W=W)
for _ in range(5):
    ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, need_dq=[False], out_dp=dp31)
for _ in range(5):
    ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, out_dp=dp31)
for _ in range(5):
    ops.synth_forward([p], [q], 4.0, soft=True, codes=True)
torch.cuda.synchronize()
print("L = 31 alg bytes fwd", rows * 128, "bwd", rows * 252, "; exact W + codes + q[code] alg bytes", rows * 137)
