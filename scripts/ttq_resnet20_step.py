"""Training step of the ResNet-20 graph with ternary (TWN / TTQ) or INQ convolutions (quantized_cnn_amd.ternary.ResNet;
no reference implementation of these quantisers exists: parity unpinned), batch 128, synthetic CIFAR-10: eager and as one
replayed HIP graph (forward, cross entropy, backward, SGD with momentum).
usage: ttq_resnet20_step.py [--modes ttq,twn,inq] [--batch 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_cnn_amd import ternary  # noqa: E402
from quantized_cnn_amd.train import CapturedStep  # noqa: E402


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def build(mode, batch):
    c = argparse.Namespace(mode="train", mode_q=mode)
    torch.manual_seed(0)
    model = ternary.ResNet(c, 3).cuda().train()
    if mode == "inq":
        for m in model.modules():
            if hasattr(m, "quantize_portion"):
                m.quantize_portion(0.5)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(batch, 3, 32, 32, device="cuda", generator=g)
    y = torch.randint(0, 10, (batch,), device="cuda", generator=g)
    return model, opt, x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="ttq,twn,inq")
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    for mode in a.modes.split(","):
        model, opt, x, y = build(mode, a.batch)

        def eager():
            loss = torch.nn.functional.cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
        l0 = float(eager())
        te = timed(eager, 20, 5)
        model, opt, x, y = build(mode, a.batch)
        step = CapturedStep(model, opt, torch.nn.functional.cross_entropy, warmup=3)
        tg = timed(lambda: step(x, y), 100, 20)
        print("ResNet-20 %s batch %d: eager %.3f ms/step %.0f img/s | one HIP graph %.3f ms/step %.0f img/s | loss %.4f -> %.4f" % (
            mode, a.batch, te * 1e3, a.batch / te, tg * 1e3, a.batch / tg, l0, float(step.loss)), flush=True)


if __name__ == "__main__":
    main()
