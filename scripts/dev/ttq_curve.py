"""Diagnostic: loss curves of the TTQ ResNet-20 step, autograd path (TrainStep) against FusedTernaryStep, batch 128, lr 1e-2."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import quantized_cnn_amd.ternary as T
from quantized_cnn_amd.train import FusedTernaryStep, TrainStep
g = torch.Generator(device="cuda").manual_seed(1000)
x = torch.randn(128, 3, 32, 32, device="cuda", generator=g); y = torch.randint(0, 10, (128,), device="cuda", generator=g)
for mode in ("ttq", "twn"):
    for kind in ("autograd", "fused", "fused_graph"):
        c = argparse.Namespace(mode="train", mode_q=mode, l2_reg=5e-4, lr=1e-2, prim_lr=1e-2, q_lr=1e-3)
        torch.manual_seed(0)
        m = T.ResNet(c, 3).cuda().train()
        step = TrainStep(m, c, T) if kind == "autograd" else FusedTernaryStep(m, c, use_graph=kind.endswith("graph"))
        ls = [float(step(x, y)) for _ in range(121)]
        print(mode, kind, " ".join("%.4f" % l for l in ls[::10]), flush=True)
