"""Diagnostic: per-parameter gradient error of the ResNet-18 @224 step (this library, and torch's own fp32 operators on the
GPU) against a float64 evaluation on the host."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_resnet18 import config, stock_forward, _float64_reference
from quantized_cnn_amd.resnet18 import ResNet18
for B in (4, 16):
    torch.manual_seed(0)
    model = ResNet18(config("inq"), num_classes=10).cuda().train()
    x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
    loss = F.cross_entropy(model(x), y); loss.backward()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    ref = F.cross_entropy(stock_forward(model, x), y); ref.backward()
    tg = {n: p.grad.clone() for n, p in model.named_parameters()}
    l64, g64 = _float64_reference(model, x, y)
    print("batch", B, "loss ours %.8f torch %.8f f64 %.8f" % (float(loss), float(ref), l64))
    for n in got:
        s = float(g64[n].abs().max())
        eo = float((got[n].cpu().double() - g64[n]).abs().max()) / s
        et = float((tg[n].cpu().double() - g64[n]).abs().max()) / s
        d = (got[n].cpu().double() - g64[n])
        l2o = float(d.norm() / g64[n].norm()); l2t = float((tg[n].cpu().double() - g64[n]).norm() / g64[n].norm())
        fr = float((d.abs() > 1e-3 * s).double().mean())
        print("  %-28s ours %.2e torch %.2e ratio %.2f | L2 ours %.2e torch %.2e | share beyond 1e-3 of max: %.4f" % (n, eo, et, eo / max(et, 1e-30), l2o, l2t, fr))
