"""Element-wise accuracy of the synthesis kernels (fast and exact arithmetic) against a float64 evaluation of the
reference formulas (adp_qtz_fan.py:563-568 and its autograd) on the GPU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantized_cnn_amd.ops as ops  # noqa: E402

torch.manual_seed(0)
rows, L = 200_000, 32
for name, qgen in (("fan-like q", lambda: (torch.rand(L, device="cuda") * 2 - 1) * 0.2),
                   ("hint-like q", lambda: torch.tensor([-(2.0 ** -k) for k in range(2, 18)] + [2.0 ** -k for k in range(17, 1, -1)], device="cuda"))):
    for beta in (1.0, 4.0):
        p = torch.rand(rows, L, device="cuda")
        q = qgen()
        dW = torch.randn(rows, device="cuda")
        p64 = p.double().requires_grad_(True)
        q64 = q.double().requires_grad_(True)
        pn = p64 / torch.sqrt((p64 ** 2).sum(-1, keepdim=True))
        e = torch.exp(beta * pn)
        W64 = (e / e.sum(-1, keepdim=True)) @ q64
        (W64 * dW.double()).sum().backward()
        # the reference's own fp32 evaluation on the GPU (torch ops), as a yardstick
        p32 = p.clone().requires_grad_(True)
        pn32 = p32 / torch.sqrt((p32 ** 2).sum(-1, keepdim=True))
        e32 = torch.exp(beta * pn32)
        W32 = (e32 / e32.sum(-1, keepdim=True)) @ q
        (W32 * dW).sum().backward()
        print("%s beta=%g" % (name, beta))
        for label, exact in (("torch fp32", None), ("fast", False), ("exact", True)):
            if exact is None:
                W, dp = W32.detach(), p32.grad
            else:
                W = ops.synth_forward([p], [q], beta, exact=exact)["W"][0]
                dp = ops.synth_backward([p], [q], beta, [dW], 0.0, exact=exact)[0][0]
            eW = (W.double() - W64.detach()).abs() / q.abs().max().double()
            rel = (dp.double() - p64.grad).abs() / p64.grad.abs().clamp_min(1e-30)
            rowrel = (dp.double() - p64.grad).abs().max(1).values / p64.grad.abs().max(1).values
            print("   %-10s W err/max|q|: max %.2e mean %.2e | dp rel: median %.2e p99 %.2e | dp err/rowmax: max %.2e mean %.2e" % (
                label, eW.max(), eW.mean(), rel.median(), rel.flatten()[::7].quantile(0.99), rowrel.max(), rowrel.mean()))
