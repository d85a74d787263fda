"""ResNet-18 @ 224 x 224 (no reference model) through the general convolution kernels: time of a training step per
quantiser mode, and the per-layer times / MFMA fractions of the convolution kernels at that batch."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantized_cnn_amd.adp_qtz_fan as fan  # noqa: E402
import quantized_cnn_amd.ops as ops  # noqa: E402
from quantized_cnn_amd.resnet18 import ResNet18  # noqa: E402

PEAK = 157.3


def ev(fn, it=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--modes", default="inq,ttq")
    ap.add_argument("--layers", action="store_true")
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    B = a.batch
    if a.layers:
        shapes = [(3, 64, 7, 2, 3, 224), (64, 64, 3, 1, 1, 56), (64, 128, 3, 2, 1, 56), (64, 128, 1, 2, 0, 56), (128, 128, 3, 1, 1, 28),
                  (128, 256, 3, 2, 1, 28), (256, 256, 3, 1, 1, 14), (256, 512, 3, 2, 1, 14), (512, 512, 3, 1, 1, 7)]
        for cin, cout, ks, s, p, H in shapes:
            x = torch.randn(B, cin, H, H, device="cuda")
            W = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
            y = ops.conv_forward(x, W, s, pad=p)
            dy = torch.randn_like(y)
            fl = 2.0 * B * cout * cin * ks * ks * y.shape[2] * y.shape[3]
            tf = ev(lambda: ops.conv_forward(x, W, s, pad=p))
            td = ev(lambda: ops.conv_dgrad(dy, W, x.shape, s, pad=p))
            tw = ev(lambda: ops.conv_wgrad(x, dy, s, ks=ks, pad=p))
            print("conv %3d->%3d k%d s%d @%3d N=%d  fwd %7.1f us %5.1f TF (%.2f)  dgrad %7.1f us %5.1f TF (%.2f)  wgrad %7.1f us %5.1f TF (%.2f)" % (
                cin, cout, ks, s, H, B, tf * 1e6, fl / tf / 1e12, fl / tf / 1e12 / PEAK, td * 1e6, fl / td / 1e12, fl / td / 1e12 / PEAK,
                tw * 1e6, fl / tw / 1e12, fl / tw / 1e12 / PEAK), flush=True)
    for mode in a.modes.split(","):
        c = argparse.Namespace(mode="train", mode_q=mode, prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=1e-4)
        torch.manual_seed(0)
        fan.beta = 1
        model = ResNet18(c, fan).cuda().train()
        if mode == "inq":
            for m in model.modules():
                if hasattr(m, "quantize_portion"):
                    m.quantize_portion(0.5)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
        x = torch.randn(B, 3, 224, 224, device="cuda")
        y = torch.randint(0, 1000, (B,), device="cuda")

        def step():
            loss = torch.nn.functional.cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
        l0 = float(step())
        t = ev(step, 5, 1)
        print("ResNet-18 %s batch %d: %.1f ms/step, %.0f img/s, loss %.4f -> %.4f; conv flops %.1f GFLOP/step = %.1f TF/s" % (
            mode, B, t * 1e3, B / t, l0, float(step()), 3 * 1.8 * B, 3 * 1.8e9 * B / t / 1e12), flush=True)
        if a.graph:  # the same step captured once and replayed as one HIP graph
            from quantized_cnn_amd.train import CapturedStep
            cs = CapturedStep(model, opt, torch.nn.functional.cross_entropy)
            tg = ev(lambda: cs(x, y), 10, 2)
            print("ResNet-18 %s batch %d as one HIP graph: %.2f ms/step, %.0f img/s, loss %.4f" % (mode, B, tg * 1e3, B / tg, float(cs.loss)), flush=True)


if __name__ == "__main__":
    main()
